// Micro-benchmark (round 2): does v_mfma_f32_16x16x4_f32 overlap with NON-packed vector work issued by
// the same wave?  Round 1 only measured v_pk_fma_f32 fillers, the one form MI355X_MICROARCH.md lists as an anti-lever
// beside MFMAs.  Every instruction here is inline asm (asm volatile keeps program order; nothing for the SLP
// vectoriser to pack).  Per MFMA: NF fillers of kind KIND.
//   KIND 0: v_fma_f32   1: v_pk_fma_f32   2: v_mov_b32 dpp quad_perm   3: v_rcp_f32   4: ds_read_b128   5: v_add_u32
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/coissue2.hip -o tools/micro/_build/coissue2 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__device__ __forceinline__ void filler(float& x, f32x2& p, f32x4& l, float a, float b, unsigned lds_addr) {
    if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(f32x2{a, b}));
    if constexpr (KIND == 2) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
    if constexpr (KIND == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
    if constexpr (KIND == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(l) : "v"(lds_addr));
    if constexpr (KIND == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));
}

template <int KIND, int NF, int NM>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    __shared__ f32x4 lds[256];
    lds[threadIdx.x] = f32x4{a, b, a, b};
    __syncthreads();
    f32x4 acc[4];
    float x[8];
    f32x2 p[8];
    f32x4 l[8];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) { x[i] = (float)threadIdx.x + i; p[i] = f32x2{x[i], 1.0f}; l[i] = f32x4{0, 0, 0, 0}; }
    const unsigned lds_addr = (unsigned)(threadIdx.x * 16);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if constexpr (NM) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < NF; ++j) filler<KIND>(x[j % 8], p[j % 8], l[j % 8], a, b, lds_addr);
        }
        if constexpr (KIND == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += x[i] + p[i][0] + p[i][1] + l[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static const char* kind_name[] = {"v_fma_f32", "v_pk_fma_f32", "v_mov_dpp", "v_rcp_f32", "ds_read_b128", "v_add_u32"};

template <int KIND, int NF, int NM>
void run(float* d, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND, NF, NM><<<blocks, 256>>>(d, 100, 1.0f, 0.5f);
    hipEventRecord(e0);
    k<KIND, NF, NM><<<blocks, 256>>>(d, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // cycles per (MFMA + its fillers) per wave at 2.4 GHz nominal: ms * 2.4e6 / (iters * 4) / waves_per_simd
    const double wps = blocks * 4.0 / 1024.0;
    const double cyc = ms * 2.4e6 / (iters * 4.0) / (wps < 1 ? 1 : wps);
    const double mf = blocks * 4.0 * iters * 4.0 * NM * 2048;
    printf("%-13s NF=%2d MFMA=%d waves/SIMD=%.0f: %8.3f ms  %6.1f cyc@2.4GHz per group  mfma %.1f TF\n", kind_name[KIND], NF, NM, wps,
           ms, cyc, mf / ms / 1e9);
    fflush(stdout);
}

template <int KIND>
void sweep(float* d, int blocks) {
    run<KIND, 4, 0>(d, blocks);
    run<KIND, 8, 0>(d, blocks);
    run<KIND, 2, 1>(d, blocks);
    run<KIND, 4, 1>(d, blocks);
    run<KIND, 6, 1>(d, blocks);
    run<KIND, 8, 1>(d, blocks);
    run<KIND, 12, 1>(d, blocks);
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 2048 * 256 * 4);
    for (int blocks : {256, 512}) {
        printf("---- %d blocks of 256 threads (%d wave(s) per SIMD) ----\n", blocks, blocks / 256);
        run<0, 0, 1>(d, blocks);
        sweep<0>(d, blocks);
        sweep<1>(d, blocks);
        sweep<2>(d, blocks);
        sweep<3>(d, blocks);
        sweep<4>(d, blocks);
        sweep<5>(d, blocks);
    }
    return 0;
}
