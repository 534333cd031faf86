// Micro-benchmark (round 2): does v_mfma_f32_16x16x32_f16 (the conv0_f16 / conv0_split matrix instruction) overlap with
// ordinary vector work - issued by the SAME wave (interleaved) or by ANOTHER wave of the same SIMD (wave-specialised)?
// coissue2.hip answered this for v_mfma_f32_16x16x4_f32 only (additive).  All inline asm, program order kept.
//   MODE 0: every wave runs  [1 MFMA + NF fillers] x 4 per iteration
//   MODE 1: waves with even (threadIdx.x / 64 / 4) ... see below: half of the waves of a SIMD run MFMAs only, the other half
//           fillers only (2 blocks per CU -> 2 waves per SIMD; block parity picks the role)
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/coissue3.hip -o tools/micro/_build/coissue3 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ void filler(float& x, float a, float b) {
    if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    if constexpr (KIND == 1) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(x));
    if constexpr (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));
}

template <int KIND, int NF, int NM, int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters, float a, float b) {
    f32x4 acc[4];
    float x[8];
    h16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (_Float16)a; bv[i] = (_Float16)b; x[i] = (float)threadIdx.x + i; }
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
    // MODE 1: waves 0..3 of the 8-wave block (one per SIMD) run MFMAs only, waves 4..7 (the second wave of each SIMD)
    // fillers only; the role branch is outside the loop
    const int role = MODE == 0 ? 2 : (int)(threadIdx.x >> 8);
    if (role == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (NM) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(av), "v"(bv));
#pragma unroll
                for (int j = 0; j < NF; ++j) filler<KIND>(x[j % 8], a, b);
            }
        }
    } else if (role == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (NM) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(av), "v"(bv));
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4 * NF; ++j) filler<KIND>(x[j % 8], a, b);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

static const char* kind_name[] = {"v_fma_f32", "v_cvt_f16_f32", "v_add_u32"};

template <int KIND, int NF, int NM, int MODE>
void run(float* d, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND, NF, NM, MODE><<<blocks, 512>>>(d, 100, 1.0f, 0.5f);
    hipEventRecord(e0);
    k<KIND, NF, NM, MODE><<<blocks, 512>>>(d, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // cycles one SIMD spends per iteration-quarter (the unit that holds 1 MFMA and NF fillers of each of its waves)
    const double cyc = ms * 2.4e6 / (iters * 4.0);
    printf("%-13s mode %d NF=%2d MFMA=%d blocks=%d: %8.3f ms  %6.1f cyc@2.4GHz per SIMD per unit\n", kind_name[KIND], MODE, NF, NM,
           blocks, ms, cyc);
    fflush(stdout);
}

template <int KIND>
void sweep(float* d) {
    printf("-- %s (256 blocks x 8 waves: 2 waves per SIMD) --\n", kind_name[KIND]);
    run<KIND, 0, 1, 0>(d, 256);   // both waves MFMA only
    run<KIND, 0, 1, 1>(d, 256);   // one wave MFMA only, the other idle
    run<KIND, 4, 0, 0>(d, 256);   // both waves fillers only
    run<KIND, 4, 0, 1>(d, 256);   // one wave fillers only
    run<KIND, 8, 0, 1>(d, 256);
    run<KIND, 2, 1, 0>(d, 256);   // interleaved in both waves
    run<KIND, 4, 1, 0>(d, 256);
    run<KIND, 8, 1, 0>(d, 256);
    run<KIND, 2, 1, 1>(d, 256);   // specialised: one wave MFMA, the other NF fillers per MFMA
    run<KIND, 4, 1, 1>(d, 256);
    run<KIND, 8, 1, 1>(d, 256);
}

int main() {
    float* d;
    hipMalloc(&d, 512 * 2048 * 4);
    sweep<0>(d);
    sweep<1>(d);
    sweep<2>(d);
    return 0;
}
