// Micro-benchmark: v_mfma_f32_16x16x4_f32 issue rate of ONE wave per SIMD as a function of the number of independent
// accumulator chains (dependent back-to-back MFMAs wait for the previous result).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NC>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    f32x4 acc[NC];
    for (int i = 0; i < NC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m % NC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m % NC], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NC>
void run(float* d, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NC><<<blocks, 256>>>(d, 100, 1.0f, 0.5f);
    (void)hipEventRecord(e0);
    k<NC><<<blocks, 256>>>(d, iters, 1.0f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("chains %d, %d waves/SIMD: %.3f ms  %.1f TF  (%.1f cycles per MFMA per wave-slot at 2.4 GHz)\n", NC, blocks / 256, ms,
           blocks * 4.0 * iters * 8.0 * 2048 / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 8.0) / (blocks / 256));
}

int main() {
    float* d;
    (void)hipMalloc(&d, 2048 * 256 * 4);
    run<1>(d, 256); run<2>(d, 256); run<3>(d, 256); run<4>(d, 256); run<8>(d, 256);
    run<1>(d, 512); run<2>(d, 512); run<4>(d, 512);
    return 0;
}
