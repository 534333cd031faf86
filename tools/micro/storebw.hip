// Micro-benchmark (round 2): what does the 1.81 GB store stream of K3 cost by itself, in the shapes a kernel can give
// it?  Output = (D=256, h=192, w=288, C=32) fp32, channel-last: a workgroup of 256 threads stores, per plane, one
// row segment of 32 pixels x 128 B = 4 KiB contiguous (one float4 per thread).
//   pattern 0  K3 today: XCD-banded decode, 4 planes per workgroup (4 x 4 KiB at plane stride), plain stores
//   pattern 1  same, nontemporal stores
//   pattern 2  8 planes per workgroup
//   pattern 3  linear: workgroup i writes 16 KiB contiguous (upper bound of a fill)
//   pattern 4  linear, persistent grid (2048 workgroups looping)
//   pattern 5  copy (read 0.9 GB + write 0.9 GB): the "measured copy peak" convention of bench.py
//   pattern 6  K3 decode with SPIN dependent FMAs per thread before the stores (compute that holds the wave slot)
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/storebw.hip -o tools/micro/_build/storebw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int D = 256, H = 192, W = 288, C = 32;
constexpr size_t PLANE = (size_t)H * W * C;  // floats

template <int DPB, bool NT, int SPIN>
__global__ void __launch_bounds__(256) k3_like(float* __restrict__ out, int tiles_per_xcd, int tiles_x, float seed) {
    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    const int dchunks = D / DPB;
    const int dc = j % dchunks; j /= dchunks;
    const int tile = xcd * tiles_per_xcd + j;
    if (tile >= tiles_x * H) return;
    const int y = tile / tiles_x, x0 = (tile - y * tiles_x) * 32;
    float v = seed + threadIdx.x;
    if constexpr (SPIN > 0) {
#pragma unroll 16
        for (int i = 0; i < SPIN; ++i) v = fmaf(v, 1.0001f, 0.5f);
    }
    const float4 r = make_float4(v, v + 1, v + 2, v + 3);
#pragma unroll
    for (int i = 0; i < DPB; ++i) {
        float4* p = reinterpret_cast<float4*>(out + (size_t)(dc * DPB + i) * PLANE + ((size_t)y * W + x0) * C) + threadIdx.x;
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        if constexpr (NT) __builtin_nontemporal_store(f32x4{r.x, r.y, r.z, r.w}, reinterpret_cast<f32x4*>(p));
        else *p = r;
    }
}

__global__ void __launch_bounds__(256) linear_fill(float4* __restrict__ out, size_t n4, float seed) {
    const float v = seed + threadIdx.x;
    const float4 r = make_float4(v, v + 1, v + 2, v + 3);
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 1024) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k * 256 < n4) out[i + k * 256] = r;
    }
}

__global__ void __launch_bounds__(256) copy_k(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 1024) {
        float4 t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = (i + k * 256 < n4) ? in[i + k * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k * 256 < n4) out[i + k * 256] = t[k];
    }
}

template <class F>
static void timeit(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    float best = 1e9f, sum = 0;
    const int n = 10;
    for (int i = 0; i < n; ++i) {
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        sum += ms;
    }
    printf("%-58s avg %.3f ms  min %.3f ms  %.0f GB/s (avg)\n", name, sum / n, best, bytes / (sum / n) / 1e6);
    fflush(stdout);
}

int main() {
    const size_t total = (size_t)D * PLANE;  // floats: 1.81 GB
    float* out;
    float* in;
    (void)hipMalloc(&out, total * 4);
    (void)hipMalloc(&in, total * 2);
    (void)hipMemset(in, 0, total * 2);
    const double bytes = (double)total * 4;
    const int tiles_x = W / 32, tiles = tiles_x * H, tpx = (tiles + 7) / 8;
    timeit("0 K3 decode, 4 planes/WG, plain", bytes, [&] { k3_like<4, false, 0><<<8 * tpx * (D / 4), 256>>>(out, tpx, tiles_x, 1.f); });
    timeit("1 K3 decode, 4 planes/WG, nontemporal", bytes, [&] { k3_like<4, true, 0><<<8 * tpx * (D / 4), 256>>>(out, tpx, tiles_x, 1.f); });
    timeit("2 K3 decode, 8 planes/WG, plain", bytes, [&] { k3_like<8, false, 0><<<8 * tpx * (D / 8), 256>>>(out, tpx, tiles_x, 1.f); });
    timeit("2b K3 decode, 16 planes/WG, plain", bytes, [&] { k3_like<16, false, 0><<<8 * tpx * (D / 16), 256>>>(out, tpx, tiles_x, 1.f); });
    timeit("2c K3 decode, 1 plane/WG, plain", bytes, [&] { k3_like<1, false, 0><<<8 * tpx * (D / 1), 256>>>(out, tpx, tiles_x, 1.f); });
    timeit("3 linear fill, 16 KiB per WG, one pass", bytes, [&] { linear_fill<<<(unsigned)((total / 4 + 1023) / 1024), 256>>>((float4*)out, total / 4, 1.f); });
    timeit("4 linear fill, persistent 2048 WGs", bytes, [&] { linear_fill<<<2048, 256>>>((float4*)out, total / 4, 1.f); });
    timeit("4b linear fill, persistent 4096 WGs", bytes, [&] { linear_fill<<<4096, 256>>>((float4*)out, total / 4, 1.f); });
    timeit("5 copy 0.9 GB -> 0.9 GB, persistent 4096 WGs", bytes, [&] { copy_k<<<4096, 256>>>((const float4*)in, (float4*)out, total / 8); });
    timeit("5b hipMemsetAsync", bytes, [&] { (void)hipMemsetAsync(out, 0, total * 4, 0); });
    timeit("6 K3 decode, 4 planes/WG + 256 dependent FMAs", bytes, [&] { k3_like<4, false, 256><<<8 * tpx * (D / 4), 256>>>(out, tpx, tiles_x, 1.f); });
    timeit("6b K3 decode, 4 planes/WG + 1024 dependent FMAs", bytes, [&] { k3_like<4, false, 1024><<<8 * tpx * (D / 4), 256>>>(out, tpx, tiles_x, 1.f); });
    timeit("6c K3 decode, 4 planes/WG + 2048 dependent FMAs", bytes, [&] { k3_like<4, false, 2048><<<8 * tpx * (D / 4), 256>>>(out, tpx, tiles_x, 1.f); });
    return 0;
}
