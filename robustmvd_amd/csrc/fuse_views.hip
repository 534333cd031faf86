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

}  // namespace mvd

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
