// Micro-benchmark: can one wave overlap v_mfma_f32_16x16x4_f32 with independent v_pk_fma_f32 on gfx950?
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/coissue.hip -o gpurun_out/coissue ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV, int NM>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    f32x4 acc[4];
    f32x2 v[8];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) v[i] = f32x2{(float)threadIdx.x, 1.0f};
    const f32x2 w = {a, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (NM) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j % 8] = __builtin_elementwise_fma(v[j % 8], w, w);
            __builtin_amdgcn_sched_group_barrier(0x008, NM ? 1 : 0, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);          // VALU
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int NM>
void run(float* d, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, NM><<<blocks, 256>>>(d, 100, 1.0f, 0.5f);
    hipEventRecord(e0);
    k<NV, NM><<<blocks, 256>>>(d, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = blocks * 4.0, mf = waves * iters * 4.0 * NM * 1024 * 2, vf = waves * iters * 4.0 * NV * 128 * 2;
    printf("NV=%2d NM=%d blocks=%d: %.3f ms  mfma %.1f TF  valu %.1f TF  total %.1f TF\n", NV, NM, blocks, ms, mf / ms / 1e9, vf / ms / 1e9,
           (mf + vf) / ms / 1e9);
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 2048 * 256 * 4);
    for (int blocks : {256 * 2, 256 * 4}) {
        run<0, 1>(d, blocks);
        run<8, 0>(d, blocks);
        run<2, 1>(d, blocks);
        run<4, 1>(d, blocks);
        run<6, 1>(d, blocks);
        run<8, 1>(d, blocks);
        run<12, 1>(d, blocks);
    }
    return 0;
}
