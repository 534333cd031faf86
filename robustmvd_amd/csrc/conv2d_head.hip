// FeatureNet's two full-resolution layers in one pass (rmvd/models/blocks/mvsnet_components.py:47-48:
// ConvBnReLU(3, 8, 3, 1, 1) -> ConvBnReLU(8, 8, 3, 1, 1)) — HBM-bound: 3 input and 8 output channels per pixel, 1.6 kFLOP between.
// As two launches the 8-channel intermediate (141 MB at 5 x 768 x 1152) is written and read once each and both layers sit at
// about half the streaming rate; here a workgroup keeps the intermediate of its 8 x 64-pixel tile (+ one halo ring) in LDS:
//   1. the 3-channel image patch (12 x 68) -> LDS, zeros outside the image;
//   2. conv0 + folded BN + ReLU on the 10 x 66 pixels the tile's conv1 needs, one pixel per lane -> LDS as two float4 planes
//      (positions outside the image are conv1's zero padding, NOT conv0 of padding);
//   3. conv1 + folded BN + ReLU, two pixels per lane four rows apart (lanes = consecutive columns: conflict-free b128 reads),
//      channel-last float4 stores.
// fp32 on the vector ALU, every product one lane of a v_pk_fma_f32 (pixel value broadcast x a pair of output channels' weights);
// the 216 + 576 weights (packed [ky][kx][cin][cout] by the caller) arrive through the scalar cache as SGPR operands, one tap's 24 / 64
// per iteration of a runtime tap loop (unrolled, the compiler issues every load up front: 900 spilled SGPRs; as broadcast LDS
// reads they made the kernel LDS-bound: 16 weight reads per 64 packed FMAs, the LDS shared by four SIMDs — 158 us).
#include "mvd_common.h"

namespace mvd {

typedef float hd2 __attribute__((ext_vector_type(2)));
typedef float hd4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) hd4* hd_c4;  // constant address space: uniform loads go through the scalar cache

constexpr int HD_TH = 8, HD_TW = 64;

struct HeadParams {
    const float* img;  // (B,3,H,W)
    const float* w0;   // [3][3][3][8]  ky, kx, cin, cout
    const float* s0;   // 8
    const float* b0;   // 8
    const float* w1;   // [3][3][8][8]
    const float* s1;
    const float* b1;
    float* y;          // (B,H,W,8)
    float* tile_max;   // optional: one float per workgroup, max |y| over the finite outputs of its tile
    int B, H, W, tiles_x, tiles_y;
};

__global__ void __launch_bounds__(256) conv2d_head_kernel(HeadParams p) {
    constexpr int IR = HD_TH + 4, IC = HD_TW + 4;  // image patch
    constexpr int MR = HD_TH + 2, MC = HD_TW + 2;  // conv0 outputs the tile's conv1 reads
    __shared__ float img[3][IR][IC];
    __shared__ __attribute__((aligned(16))) hd4 mid[2][MR][MC];  // channels 0-3 / 4-7

    const int tid = threadIdx.x;
    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x; bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const int r0 = ty * HD_TH, c0 = tx * HD_TW;
    const int H = p.H, W = p.W;

    // ---- 1: image patch ----
    const float* __restrict__ im = p.img + (size_t)b * 3 * H * W;
    for (int e = tid; e < 3 * IR * IC; e += 256) {
        const int ch = e / (IR * IC), rem = e - ch * (IR * IC);
        const int r = rem / IC, c = rem - r * IC;
        const int gr = r0 - 2 + r, gc = c0 - 2 + c;
        const bool in = gr >= 0 && gr < H && gc >= 0 && gc < W;
        (&img[0][0][0])[e] = in ? im[((size_t)ch * H + gr) * W + gc] : 0.0f;
    }
    __syncthreads();

    // ---- 2: conv0 on the tile + halo ----
    for (int e = tid; e < MR * MC; e += 256) {
        const int mr = e / MC, mc = e - mr * MC;
        hd2 acc[4] = {hd2{0.f, 0.f}, hd2{0.f, 0.f}, hd2{0.f, 0.f}, hd2{0.f, 0.f}};
#pragma unroll 1  // a runtime loop over the taps (see the header)
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - 3 * ky;
            const hd_c4 wt = (hd_c4)(unsigned long long)(p.w0 + t * 24);  // the tap's 24 weights: scalar loads, SGPR operands
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float x = img[ci][mr + ky][mc + kx];
                const hd4 wa = wt[ci * 2], wb = wt[ci * 2 + 1];
                const hd2 xx = {x, x};
                acc[0] = __builtin_elementwise_fma(xx, wa.xy, acc[0]);
                acc[1] = __builtin_elementwise_fma(xx, wa.zw, acc[1]);
                acc[2] = __builtin_elementwise_fma(xx, wb.xy, acc[2]);
                acc[3] = __builtin_elementwise_fma(xx, wb.zw, acc[3]);
            }
        }
        const int gr = r0 - 1 + mr, gc = c0 - 1 + mc;
        const bool in = gr >= 0 && gr < H && gc >= 0 && gc < W;
        const hd_c4 sp = (hd_c4)(unsigned long long)p.s0, bp = (hd_c4)(unsigned long long)p.b0;
        const hd4 sa = sp[0], sb = sp[1], ba = bp[0], bb = bp[1];
        hd4 va = {fmaf(acc[0].x, sa.x, ba.x), fmaf(acc[0].y, sa.y, ba.y), fmaf(acc[1].x, sa.z, ba.z), fmaf(acc[1].y, sa.w, ba.w)};
        hd4 vb = {fmaf(acc[2].x, sb.x, bb.x), fmaf(acc[2].y, sb.y, bb.y), fmaf(acc[3].x, sb.z, bb.z), fmaf(acc[3].y, sb.w, bb.w)};
        const hd4 zero = {0.f, 0.f, 0.f, 0.f};
        va = in ? __builtin_elementwise_max(va, zero) : zero;
        vb = in ? __builtin_elementwise_max(vb, zero) : zero;
        mid[0][mr][mc] = va;
        mid[1][mr][mc] = vb;
    }
    __syncthreads();

    // ---- 3: conv1, pixels (row, col) and (row + 4, col) ----
    const int row = tid >> 6, col = tid & 63;
    hd2 accA[4] = {hd2{0.f, 0.f}, hd2{0.f, 0.f}, hd2{0.f, 0.f}, hd2{0.f, 0.f}};
    hd2 accB[4] = {hd2{0.f, 0.f}, hd2{0.f, 0.f}, hd2{0.f, 0.f}, hd2{0.f, 0.f}};
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t - 3 * ky;
        const hd4 xa0 = mid[0][row + ky][col + kx], xa1 = mid[1][row + ky][col + kx];
        const hd4 xb0 = mid[0][row + 4 + ky][col + kx], xb1 = mid[1][row + 4 + ky][col + kx];
        const hd_c4 wt = (hd_c4)(unsigned long long)(p.w1 + t * 64);  // the tap's 64 weights: scalar loads, SGPR operands
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const float xa = ci < 4 ? xa0[ci & 3] : xa1[ci & 3];
            const float xb = ci < 4 ? xb0[ci & 3] : xb1[ci & 3];
            const hd4 wa = wt[ci * 2], wb = wt[ci * 2 + 1];
            const hd2 xxa = {xa, xa}, xxb = {xb, xb};
            accA[0] = __builtin_elementwise_fma(xxa, wa.xy, accA[0]);
            accA[1] = __builtin_elementwise_fma(xxa, wa.zw, accA[1]);
            accA[2] = __builtin_elementwise_fma(xxa, wb.xy, accA[2]);
            accA[3] = __builtin_elementwise_fma(xxa, wb.zw, accA[3]);
            accB[0] = __builtin_elementwise_fma(xxb, wa.xy, accB[0]);
            accB[1] = __builtin_elementwise_fma(xxb, wa.zw, accB[1]);
            accB[2] = __builtin_elementwise_fma(xxb, wb.xy, accB[2]);
            accB[3] = __builtin_elementwise_fma(xxb, wb.zw, accB[3]);
        }
    }
    const hd_c4 sp = (hd_c4)(unsigned long long)p.s1, bp = (hd_c4)(unsigned long long)p.b1;
    const hd4 sa = sp[0], sb = sp[1], ba = bp[0], bb = bp[1];
    const hd4 zero = {0.f, 0.f, 0.f, 0.f};
    const int gc = c0 + col;
    float m = 0.0f;
    auto store = [&](const hd2 (&a)[4], int gr) {
        if (gr >= H || gc >= W) return;
        hd4 va = {fmaf(a[0].x, sa.x, ba.x), fmaf(a[0].y, sa.y, ba.y), fmaf(a[1].x, sa.z, ba.z), fmaf(a[1].y, sa.w, ba.w)};
        hd4 vb = {fmaf(a[2].x, sb.x, bb.x), fmaf(a[2].y, sb.y, bb.y), fmaf(a[3].x, sb.z, bb.z), fmaf(a[3].y, sb.w, bb.w)};
        va = __builtin_elementwise_max(va, zero);
        vb = __builtin_elementwise_max(vb, zero);
        hd4* o = reinterpret_cast<hd4*>(p.y + (((size_t)b * H + gr) * W + gc) * 8);
        o[0] = va;
        o[1] = vb;
#pragma unroll
        for (int k = 0; k < 4; ++k) m = fmaxf(m, fmaxf(finite_abs_or_zero(va[k]), finite_abs_or_zero(vb[k])));
    };
    store(accA, r0 + row);
    store(accB, r0 + row + 4);
    if (p.tile_max) {  // per-tile maxima with plain stores (a second tiny pass reduces them): no atomics on one address
        __shared__ float wmax[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((tid & 63) == 0) wmax[tid >> 6] = m;
        __syncthreads();
        if (tid == 0) p.tile_max[blockIdx.x] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    }
}

}  // namespace mvd

extern "C" {

size_t mvd_conv2d_head_tile_count(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)((W + mvd::HD_TW - 1) / mvd::HD_TW) * ((H + mvd::HD_TH - 1) / mvd::HD_TH) * B;
}

int mvd_conv2d_head_f32(const float* image, const float* w0, const float* scale0, const float* shift0, const float* w1,
                        const float* scale1, const float* shift1, float* y, float* tile_absmax, int B, int H, int W,
                        mvd_stream_t stream) {
    using namespace mvd;
    MVD_REQUIRE(image && w0 && scale0 && shift0 && w1 && scale1 && shift1 && y, "conv2d_head: NULL argument");
    MVD_REQUIRE(B > 0 && H > 0 && W > 0, "conv2d_head: non-positive dimension");
    MVD_REQUIRE((((uintptr_t)w0 | (uintptr_t)w1 | (uintptr_t)scale0 | (uintptr_t)shift0 | (uintptr_t)scale1 | (uintptr_t)shift1 | (uintptr_t)y) & 15) == 0,
                "conv2d_head: weights, scales, shifts and y must be 16-byte aligned");
    HeadParams p{};
    p.img = image; p.w0 = w0; p.s0 = scale0; p.b0 = shift0; p.w1 = w1; p.s1 = scale1; p.b1 = shift1; p.y = y; p.tile_max = tile_absmax;
    p.B = B; p.H = H; p.W = W;
    p.tiles_x = (W + HD_TW - 1) / HD_TW;
    p.tiles_y = (H + HD_TH - 1) / HD_TH;
    const long long nblk = (long long)p.tiles_x * p.tiles_y * B;
    MVD_REQUIRE(nblk <= 0x7fffffffLL, "conv2d_head: %lld workgroups exceed the grid limit", nblk);
    hipLaunchKernelGGL(conv2d_head_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status("conv2d_head");
}
}
