// K3 — fronto-parallel homography warp of V source feature maps into the key frustum at D depth
// hypotheses + variance aggregation with the key features, in one pass (Path B).
// Replaces homo_warp (rmvd/models/blocks/utils.py:222-268) and the sum / sum-of-squares / variance
// arithmetic of MVSNet.forward (rmvd/models/mvsnet.py:124-136).
//
// HBM-bound by construction: the only large tensor is the variance volume, written exactly once
// (algorithmic bytes 4*((V+1)*C*h*w + C*D*h*w) per batch element); the per-view warped volumes of the
// reference are never materialised.  Feature maps are first repacked channel-last ((h,w,C), 128 B per
// pixel at C=32) so that one bilinear tap of one pixel is a single full cache line shared by C/4 lanes.
//
// Thread mapping: C/4 lanes per output pixel (each lane owns 4 channels = one 16-B load per tap),
// 256/(C/4) consecutive x pixels per workgroup, one (b, d, y) row segment per workgroup.
#include "mvd_common.h"

namespace mvd {

struct WarpParams {
    ViewPtrs src;           // V x (B,h,w,C) channel-last source features
    ViewPtrs proj;          // V x (B,4,4) source projection matrices
    const float* key;       // (B,h,w,C) channel-last key features (unused when WARP_ONLY)
    const float* key_proj_inv;  // (B,4,4)
    const float* depth;     // (B,D)
    float* out;
    int B, D, h, w, V;
    int layout;             // MVD_LAYOUT_*
};

// row i of (src_proj @ key_proj_inv)[:3,:4] as an fmaf chain over k (what a K=4 sgemm does)
__device__ __forceinline__ void transform_rows(const float* __restrict__ P, const float* __restrict__ Q, float M[12]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = P[i * 4 + 0] * Q[0 * 4 + j];
            acc = fmaf(P[i * 4 + 1], Q[1 * 4 + j], acc);
            acc = fmaf(P[i * 4 + 2], Q[2 * 4 + j], acc);
            acc = fmaf(P[i * 4 + 3], Q[3 * 4 + j], acc);
            M[i * 4 + j] = acc;
        }
}

template <int LPP, bool WARP_ONLY>
__global__ void __launch_bounds__(256) warp_variance_kernel(WarpParams p) {
    constexpr int PPB = 256 / LPP;  // pixels per block
    constexpr int C = LPP * 4;
    __shared__ float stage[WARP_ONLY ? C * (PPB + 1) : C * (PPB + 1)];

    const int tid = threadIdx.x;
    const int q = tid % LPP;    // channel quad
    const int px = tid / LPP;   // pixel within the block
    const int x = blockIdx.x * PPB + px;
    const int y = blockIdx.y;
    const int b = blockIdx.z / p.D;
    const int d = blockIdx.z - b * p.D;
    const int h = p.h, w = p.w;
    const bool active = x < w;
    const int xc = active ? x : w - 1;

    const float depth = p.depth[b * p.D + d];
    const float gx = (float)xc * depth, gy = (float)y * depth;  // ref_grid * depth_values (utils.py:246)
    const float fw = (float)w, fh = (float)h;
    const float half_w = (float)(w - 1) / 2.0f, half_h = (float)(h - 1) / 2.0f;

    float4 s1, s2;
    if constexpr (!WARP_ONLY) {
        const float4 k = *reinterpret_cast<const float4*>(p.key + (((size_t)b * h + y) * w + xc) * C + q * 4);
        s1 = k;
        s2 = make_float4(k.x * k.x, k.y * k.y, k.z * k.z, k.w * k.w);
    } else {
        s1 = make_float4(0, 0, 0, 0);
        s2 = s1;
    }

    for (int v = 0; v < p.V; ++v) {
        float M[12];
        transform_rows(p.proj.p[v] + b * 16, p.key_proj_inv + b * 16, M);
        // R @ (x*d, y*d, d) + T, then perspective divide (utils.py:249-252)
        const float X = fmaf(M[2], depth, fmaf(M[1], gy, M[0] * gx)) + M[3];
        const float Y = fmaf(M[6], depth, fmaf(M[5], gy, M[4] * gx)) + M[7];
        const float Z = fmaf(M[10], depth, fmaf(M[9], gy, M[8] * gx)) + M[11];
        const float nx = (X / Z) / half_w - 1.0f;  // utils.py:256-257
        const float ny = (Y / Z) / half_h - 1.0f;
        const Taps t = bilinear_taps(unnormalize_coord(nx, fw), unnormalize_coord(ny, fh), h, w);
        const float* base = p.src.p[v] + (size_t)b * h * w * C + q * 4;
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 f = *reinterpret_cast<const float4*>(base + (size_t)t.off[k] * C);
            acc.x = fmaf(f.x, t.w[k], acc.x);
            acc.y = fmaf(f.y, t.w[k], acc.y);
            acc.z = fmaf(f.z, t.w[k], acc.z);
            acc.w = fmaf(f.w, t.w[k], acc.w);
        }
        s1.x += acc.x; s1.y += acc.y; s1.z += acc.z; s1.w += acc.w;
        s2.x = fmaf(acc.x, acc.x, s2.x); s2.y = fmaf(acc.y, acc.y, s2.y);
        s2.z = fmaf(acc.z, acc.z, s2.z); s2.w = fmaf(acc.w, acc.w, s2.w);
    }

    float4 r;
    if constexpr (WARP_ONLY) {
        r = s1;
    } else {
        const float nv = (float)(p.V + 1);  // mvsnet.py:135: sq/V - (sum/V)^2, V counting the key view
        const float mx = s1.x / nv, my = s1.y / nv, mz = s1.z / nv, mw = s1.w / nv;
        r = make_float4(s2.x / nv - mx * mx, s2.y / nv - my * my, s2.z / nv - mz * mz, s2.w / nv - mw * mw);
    }

    if (p.layout == MVD_LAYOUT_NDHWC) {
        if (active)
            *reinterpret_cast<float4*>(p.out + ((((size_t)b * p.D + d) * h + y) * w + x) * C + q * 4) = r;
        return;
    }
    // NCDHW: transpose the (pixel, channel) tile through LDS so every channel row is written as
    // PPB consecutive floats.
    stage[(q * 4 + 0) * (PPB + 1) + px] = r.x;
    stage[(q * 4 + 1) * (PPB + 1) + px] = r.y;
    stage[(q * 4 + 2) * (PPB + 1) + px] = r.z;
    stage[(q * 4 + 3) * (PPB + 1) + px] = r.w;
    __syncthreads();
    const size_t plane = (size_t)h * w;
#pragma unroll
    for (int i = 0; i < C * PPB / 256; ++i) {
        const int e = tid + i * 256;
        const int c = e / PPB, xx = e % PPB;
        const int gxp = blockIdx.x * PPB + xx;
        if (gxp < w)
            p.out[(((size_t)b * C + c) * p.D + d) * plane + (size_t)y * w + gxp] = stage[c * (PPB + 1) + xx];
    }
}

template <bool WARP_ONLY>
static int launch_warp(const WarpParams& p, int C, hipStream_t st) {
    const int lpp = C / 4;
    const int ppb = 256 / lpp;
    dim3 grid((unsigned)((p.w + ppb - 1) / ppb), (unsigned)p.h, (unsigned)(p.B * p.D));
    timing_begin(st);
    switch (lpp) {
#define MVD_CASE(L)                                                                                  \
    case L:                                                                                          \
        hipLaunchKernelGGL((warp_variance_kernel<L, WARP_ONLY>), grid, dim3(256), 0, st, p);        \
        break;
        MVD_CASE(1) MVD_CASE(2) MVD_CASE(4) MVD_CASE(8) MVD_CASE(16)
#undef MVD_CASE
        default:
            set_error("warp_variance: C=%d unsupported (need 4, 8, 16, 32 or 64)", C);
            return MVD_ERR_INVALID_ARG;
    }
    timing_end(st);
    return launch_status("warp_variance");
}

int transpose_launch(const float* src, float* dst, int N, long long rows, long long cols, hipStream_t st);

}  // namespace mvd

extern "C" {

size_t mvd_warp_variance_workspace_bytes(int B, int C, int h, int w, int V) {
    if (B <= 0 || C <= 0 || h <= 0 || w <= 0 || V < 0) return 0;
    return (size_t)(V + 1) * mvd::align_up((size_t)B * C * h * w * sizeof(float), 256);
}

int mvd_warp_variance_f32(const float* key_feat, const float* const* src_feat, const float* const* src_proj,
                          const float* key_proj_inv, const float* depth_values, int B, int C, int D, int h, int w,
                          int V, float* var_out, int out_layout, void* workspace, size_t workspace_bytes,
                          mvd_stream_t stream) {
    MVD_REQUIRE(key_feat && src_feat && src_proj && key_proj_inv && depth_values && var_out,
                "warp_variance: NULL argument");
    MVD_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "warp_variance: non-positive dimension");
    MVD_REQUIRE(V >= 1 && V <= MVD_MAX_VIEWS, "warp_variance: V=%d outside 1..%d", V, MVD_MAX_VIEWS);
    MVD_REQUIRE(h <= 65535 && (long long)B * D <= 65535, "warp_variance: h or B*D exceeds 65535");
    MVD_REQUIRE(out_layout == MVD_LAYOUT_NCDHW || out_layout == MVD_LAYOUT_NDHWC, "warp_variance: bad layout");
    const size_t need = mvd_warp_variance_workspace_bytes(B, C, h, w, V);
    if (!workspace || workspace_bytes < need) {
        mvd::set_error("warp_variance: workspace %zu B < required %zu B", workspace_bytes, need);
        return MVD_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t per = mvd::align_up((size_t)B * C * h * w * sizeof(float), 256) / sizeof(float);
    float* ws = (float*)workspace;
    mvd::WarpParams p{};
    int rc = mvd::transpose_launch(key_feat, ws, B, C, (long long)h * w, st);
    if (rc) return rc;
    p.key = ws;
    for (int v = 0; v < V; ++v) {
        MVD_REQUIRE(src_feat[v] && src_proj[v], "warp_variance: NULL view %d", v);
        float* dst = ws + (size_t)(v + 1) * per;
        rc = mvd::transpose_launch(src_feat[v], dst, B, C, (long long)h * w, st);
        if (rc) return rc;
        p.src.p[v] = dst;
        p.proj.p[v] = src_proj[v];
    }
    p.key_proj_inv = key_proj_inv;
    p.depth = depth_values;
    p.out = var_out;
    p.B = B; p.D = D; p.h = h; p.w = w; p.V = V;
    p.layout = out_layout;
    return mvd::launch_warp<false>(p, C, st);
}

int mvd_homo_warp_f32(const float* src_feat, const float* src_proj, const float* key_proj_inv,
                      const float* depth_values, int B, int C, int D, int h, int w, float* warped_out,
                      void* workspace, size_t workspace_bytes, mvd_stream_t stream) {
    MVD_REQUIRE(src_feat && src_proj && key_proj_inv && depth_values && warped_out, "homo_warp: NULL argument");
    MVD_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "homo_warp: non-positive dimension");
    MVD_REQUIRE(h <= 65535 && (long long)B * D <= 65535, "homo_warp: h or B*D exceeds 65535");
    const size_t need = mvd_warp_variance_workspace_bytes(B, C, h, w, 0);
    if (!workspace || workspace_bytes < need) {
        mvd::set_error("homo_warp: workspace %zu B < required %zu B", workspace_bytes, need);
        return MVD_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    int rc = mvd::transpose_launch(src_feat, ws, B, C, (long long)h * w, st);
    if (rc) return rc;
    mvd::WarpParams p{};
    p.src.p[0] = ws;
    p.proj.p[0] = src_proj;
    p.key = nullptr;
    p.key_proj_inv = key_proj_inv;
    p.depth = depth_values;
    p.out = warped_out;
    p.B = B; p.D = D; p.h = h; p.w = w; p.V = 1;
    p.layout = MVD_LAYOUT_NCDHW;
    return mvd::launch_warp<true>(p, C, st);
}
}
