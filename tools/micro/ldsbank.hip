// Micro-benchmark (round 2): LDS bank-conflict cost of the fragment-read patterns the conv kernels use.
// lane l reads WIDTH bytes at  (l % 16) * STRIDE + (l / 16) * QOFF  (+ optional XOR swizzle of the 16-byte chunk index by
// ((l % 16) >> SW_SHIFT) & SW_MASK): l % 16 = voxel, l / 16 = channel chunk, as in conv3d.hip / conv2d.hip / conv3d_f16.hip.
// Prints cycles per wave-instruction at 4 waves per CU (one per SIMD) and at 8.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/ldsbank.hip -o tools/micro/_build/ldsbank ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int WIDTH>
__global__ void __launch_bounds__(512) k(float* out, int iters, int stride, int qoff, int swm, int sws) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    const int l = threadIdx.x & 63, v = l & 15, q = l >> 4;
    unsigned a = (unsigned)(v * stride);
    if (swm) a += (unsigned)(((q ^ ((v >> sws) & swm)) * qoff));
    else a += (unsigned)(q * qoff);
    a &= 0xffff & ~(WIDTH - 1);
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (WIDTH == 16) {
                f32x4 t;
                asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(a));
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc += t;
            } else {
                f32x2 t;
                asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(a));
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc.x += t.x; acc.y += t.y;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int WIDTH>
void run(float* d, int threads, int stride, int qoff, int swm, int sws) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<WIDTH><<<256, threads, 65536>>>(d, 10, stride, qoff, swm, sws);
    (void)hipEventRecord(e0);
    k<WIDTH><<<256, threads, 65536>>>(d, iters, stride, qoff, swm, sws);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 2.4e6 / (iters * 8.0) / (threads / 64);  // CU cycles (2.4 GHz nominal) per wave-instruction
    printf("b%-3d waves/CU %d stride %4d qoff %3d swz(mask %d, shift %d): %6.2f cyc/wave-instr  (%5.1f B/clk/CU)\n", WIDTH * 8,
           threads / 64, stride, qoff, swm, sws, per, 64.0 * WIDTH / per);
}

int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 512 * 4);
    (void)hipFuncSetAttribute((const void*)k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int threads : {256, 512}) {
        printf("---- ds_read_b128, lane = (voxel l%%16) * stride + (chunk l/16) * 16 ----\n");
        for (int s : {16, 32, 48, 64, 80, 96, 112, 128, 144, 160, 192, 208, 256, 272, 288, 320}) run<16>(d, threads, s, 16, 0, 0);
        printf("---- ds_read_b128 with the chunk XOR-swizzled by (voxel >> 1) & 3 (conv3d_f16) / voxel & 7 (conv0 PAIR, stride 256) ----\n");
        run<16>(d, threads, 64, 16, 3, 1);
        run<16>(d, threads, 128, 16, 3, 1);
        run<16>(d, threads, 256, 16, 7, 0);
        run<16>(d, threads, 256, 16, 3, 0);
        printf("---- ds_read_b64, lane = voxel * stride + chunk * 8 ----\n");
        for (int s : {8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160}) run<8>(d, threads, s, 8, 0, 0);
    }
    return 0;
}
