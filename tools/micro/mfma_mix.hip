// Micro-benchmark: what do a few integer VALU ops / LDS reads sprinkled between v_mfma_f32_16x16x4_f32 cost on gfx950?
// One wave per SIMD (1024 workgroups of 64 threads... here 256-thread blocks, 1 block per CU via LDS size).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// per group of 4 MFMAs: NA v_add_u32 (each separated from the others by MFMAs when SPREAD) and ND ds_read_b128
template <int NA, int ND, bool SPREAD>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b, int inc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    unsigned addr[4] = {threadIdx.x * 16u, threadIdx.x * 16u + 4096u, threadIdx.x * 16u + 8192u, threadIdx.x * 16u + 12288u};
    f32x4 r[4] = {f32x4{a, a, a, a}, f32x4{b, b, b, b}, f32x4{a, b, a, b}, f32x4{b, a, b, a}};
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = a;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(r[m][0], r[(m + 1) & 3][1], acc[m & 1], 0, 0, 0);
            if (SPREAD) {
                if (m < NA) addr[m] = (addr[m] + inc) & 0x7ff0;
                if (m < ND) r[m] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds) + addr[m]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!SPREAD) {
#pragma unroll
            for (int m = 0; m < NA; ++m) addr[m] = (addr[m] + inc) & 0x7ff0;
#pragma unroll
            for (int m = 0; m < ND; ++m) r[m] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds) + addr[m]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 4; ++i) s += r[i][0] + addr[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NA, int ND, bool SPREAD>
void run(float* d) {
    const int iters = 20000, blocks = 256;  // one workgroup (4 waves) per CU: one wave per SIMD, like conv0
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NA, ND, SPREAD><<<blocks, 256, 96 * 1024>>>(d, 100, 1.0f, 0.5f, 16);
    (void)hipEventRecord(e0);
    k<NA, ND, SPREAD><<<blocks, 256, 96 * 1024>>>(d, iters, 1.0f, 0.5f, 16);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mf = blocks * 4.0 * iters * 4.0 * 1024 * 2;
    printf("adds %d  ds_reads %d  %s: %.3f ms  mfma %.1f TF  (%.1f cycles per MFMA at 2.4 GHz)\n", NA, ND, SPREAD ? "spread " : "grouped", ms,
           mf / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 4.0));
}

int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 1024 * 4);
    (void)hipFuncSetAttribute((const void*)k<0, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
#define RUN(A, D, S) (void)hipFuncSetAttribute((const void*)k<A, D, S>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); run<A, D, S>(d);
    RUN(0, 0, true) RUN(1, 0, true) RUN(2, 0, true) RUN(4, 0, true) RUN(2, 0, false) RUN(4, 0, false)
    RUN(0, 1, true) RUN(0, 2, true) RUN(0, 4, true) RUN(0, 2, false) RUN(0, 4, false)
    RUN(2, 2, true) RUN(2, 2, false) RUN(4, 4, true) RUN(4, 4, false)
    return 0;
}
