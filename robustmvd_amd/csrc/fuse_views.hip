// K2 — learned multi-view fusion arithmetic (Path A): softmax of the per-view score maps over the
// view axis (+1e-9), mask-weighted mean of the correlation volumes, fused mask.
// Replaces rmvd/models/blocks/learned_fusion.py:32-48.  Pure streaming: reads 2V volumes, writes 2.
#include "mvd_common.h"

namespace mvd {

struct FuseParams {
    ViewPtrs corr, mask, score;
    float* fused;
    float* fmask;
    long long hw;
    int S, V;
};

__global__ void __launch_bounds__(256) fuse_views_kernel(FuseParams p) {
    const int n = blockIdx.z;
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= p.hw) return;
    // softmax over views of the (N,1,h,w) scores, shared by all S planes of this pixel
    float wv[MVD_MAX_VIEWS];
    float m = -INFINITY;
    for (int v = 0; v < p.V; ++v) {
        wv[v] = p.score.p[v][(long long)n * p.hw + pix];
        m = fmaxf(m, wv[v]);
    }
    float se = 0.f;
    for (int v = 0; v < p.V; ++v) {
        wv[v] = expf(wv[v] - m);
        se += wv[v];
    }
    for (int v = 0; v < p.V; ++v) wv[v] = wv[v] / se + 1e-9f;  // learned_fusion.py:33

    const int s0 = blockIdx.y * 16;
    const int s1 = min(s0 + 16, p.S);
    for (int s = s0; s < s1; ++s) {
        const long long o = ((long long)n * p.S + s) * p.hw + pix;
        float wsum = 0.f, csum = 0.f;
        for (int v = 0; v < p.V; ++v) {
            const float vw = wv[v] * p.mask.p[v][o];  // :37-40
            wsum += vw;
            csum += p.corr.p[v][o] * vw;              // :44-46
        }
        const float fm = (wsum != 0.f) ? 1.f : 0.f;   // :42
        p.fused[o] = csum / (wsum + 1e-9f) * fm;      // :47
        p.fmask[o] = fm;
    }
}

// The same arithmetic on pixel-major volumes (N,h,w,S) (mvd_sweep_corr_nhwc_f32's output): one thread = 4 planes of one pixel.
// fused goes to a channel slice of the cost-volume encoder's input buffer; max |fused| by one atomic per workgroup.
struct FuseNhwcParams {
    ViewPtrs corr, mask, score;
    float* fused;
    float* fmask;   // optional
    float* amax;    // optional
    long long npix; // N h w
    int S, V, in_ps, out_ps;
};

__global__ void __launch_bounds__(256) fuse_views_nhwc_kernel(FuseNhwcParams p) {
    __shared__ float wmax[4];
    const int s4n = p.S / 4;
    float am = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.npix * s4n; i += (long long)gridDim.x * 256) {
        const long long pix = i / s4n;
        const int s = (int)(i % s4n) * 4;
        float wv[MVD_MAX_VIEWS];
        float m = -INFINITY;
        for (int v = 0; v < p.V; ++v) {
            wv[v] = p.score.p[v][pix];
            m = fmaxf(m, wv[v]);
        }
        float se = 0.f;
        for (int v = 0; v < p.V; ++v) {
            wv[v] = expf(wv[v] - m);
            se += wv[v];
        }
        for (int v = 0; v < p.V; ++v) wv[v] = wv[v] / se + 1e-9f;  // learned_fusion.py:33
        float wsum[4] = {0.f, 0.f, 0.f, 0.f}, csum[4] = {0.f, 0.f, 0.f, 0.f};
        for (int v = 0; v < p.V; ++v) {
            const float4 mk = *reinterpret_cast<const float4*>(p.mask.p[v] + pix * p.in_ps + s);
            const float4 co = *reinterpret_cast<const float4*>(p.corr.p[v] + pix * p.in_ps + s);
            const float mks[4] = {mk.x, mk.y, mk.z, mk.w}, cos_[4] = {co.x, co.y, co.z, co.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float vw = wv[v] * mks[k];  // :37-40
                wsum[k] += vw;
                csum[k] += cos_[k] * vw;          // :44-46
            }
        }
        float f[4], fm[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            fm[k] = (wsum[k] != 0.f) ? 1.f : 0.f;        // :42
            f[k] = csum[k] / (wsum[k] + 1e-9f) * fm[k];  // :47
            am = fmaxf(am, finite_abs_or_zero(f[k]));
        }
        *reinterpret_cast<float4*>(p.fused + pix * p.out_ps + s) = make_float4(f[0], f[1], f[2], f[3]);
        if (p.fmask) *reinterpret_cast<float4*>(p.fmask + pix * p.out_ps + s) = make_float4(fm[0], fm[1], fm[2], fm[3]);
    }
    if (p.amax) {
        for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = am;
        __syncthreads();
        if (threadIdx.x == 0) {
            am = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            raise_absmax(p.amax, am);
        }
    }
}

}  // namespace mvd

extern "C" int mvd_fuse_views_nhwc_f32(const float* const* corr, const float* const* mask, const float* const* score, int N, int S, int h,
                                       int w, int V, int in_pixel_stride, float* fused, float* fused_mask, int out_pixel_stride,
                                       float* fused_absmax, mvd_stream_t stream) {
    MVD_REQUIRE(corr && mask && score && fused, "fuse_views_nhwc: NULL argument");
    MVD_REQUIRE(N > 0 && S > 0 && h > 0 && w > 0 && S % 4 == 0, "fuse_views_nhwc: bad dimension (S a multiple of 4)");
    MVD_REQUIRE(V >= 2 && V <= MVD_MAX_VIEWS, "fuse_views_nhwc: V=%d outside 2..%d (V=1 is a pass-through)", V, MVD_MAX_VIEWS);
    MVD_REQUIRE(in_pixel_stride >= S && out_pixel_stride >= S && in_pixel_stride % 4 == 0 && out_pixel_stride % 4 == 0,
                "fuse_views_nhwc: pixel strides %d / %d (multiples of 4, at least S)", in_pixel_stride, out_pixel_stride);
    mvd::FuseNhwcParams p{};
    for (int v = 0; v < V; ++v) {
        MVD_REQUIRE(corr[v] && mask[v] && score[v], "fuse_views_nhwc: NULL view %d", v);
        p.corr.p[v] = corr[v];
        p.mask.p[v] = mask[v];
        p.score.p[v] = score[v];
    }
    p.fused = fused; p.fmask = fused_mask; p.amax = fused_absmax;
    p.npix = (long long)N * h * w;
    p.S = S; p.V = V; p.in_ps = in_pixel_stride; p.out_ps = out_pixel_stride;
    const long long n = p.npix * (S / 4);
    hipLaunchKernelGGL(mvd::fuse_views_nhwc_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0,
                       (hipStream_t)stream, p);
    return mvd::launch_status("fuse_views_nhwc");
}

extern "C" int mvd_fuse_views_f32(const float* const* corr, const float* const* mask, const float* const* score,
                                  int N, int S, int h, int w, int V, float* fused, float* fused_mask,
                                  mvd_stream_t stream) {
    MVD_REQUIRE(corr && mask && score && fused && fused_mask, "fuse_views: NULL argument");
    MVD_REQUIRE(N > 0 && S > 0 && h > 0 && w > 0 && N <= 65535, "fuse_views: bad dimension");
    MVD_REQUIRE(V >= 2 && V <= MVD_MAX_VIEWS, "fuse_views: V=%d outside 2..%d (V=1 is a pass-through)", V,
                MVD_MAX_VIEWS);
    mvd::FuseParams p{};
    for (int v = 0; v < V; ++v) {
        MVD_REQUIRE(corr[v] && mask[v] && score[v], "fuse_views: NULL view %d", v);
        p.corr.p[v] = corr[v];
        p.mask.p[v] = mask[v];
        p.score.p[v] = score[v];
    }
    p.fused = fused;
    p.fmask = fused_mask;
    p.hw = (long long)h * w;
    p.S = S;
    p.V = V;
    dim3 grid((unsigned)((p.hw + 255) / 256), (unsigned)((S + 15) / 16), (unsigned)N);
    hipLaunchKernelGGL(mvd::fuse_views_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return mvd::launch_status("fuse_views");
}
