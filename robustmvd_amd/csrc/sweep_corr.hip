// K1 — inverse-depth plane sweep with dot-product correlation (Path A, robust_mvd).
// Replaces PlanesweepCorrelation.forward (rmvd/models/blocks/planesweep_corr.py:396-521):
// epipolar coefficients (:228-300), sampling grids with non-finite replacement (:333-349), visibility
// mask (:489-512) and TorchCorr (:152-195, warp() :49-104) for all V source views in one launch.
//
// The reference materialises the (h*w)x(hs*ws) all-pairs matrix per view (764 MB at 96x144) and
// interpolates it; here each (key pixel, plane) sample gathers its 4 bilinear taps straight from a
// channel-last copy of the source features (1 KB per pixel at C=256) and reduces the C-long dot
// product inside a 16-lane group, so neither the all-pairs matrix nor the sampling grids ever exist
// in memory.  HBM traffic is the feature maps in and 2*V*S*h*w floats out; the kernel is bound by
// L1 gather bandwidth and vector FMA, not HBM.
//
// Mapping: a wave = 4 groups of 16 lanes; group g works on plane s = 4*i + g of ONE key pixel, its 16
// lanes each own C/16 channels (C/64 float4 chunks, chunk j covering channels 64*j + 4*lane..+3, so a
// group's load of one chunk is 256 contiguous bytes).  A workgroup (4 waves) covers 16 consecutive key
// pixels of one row for one view; results are staged in LDS and written as 64-B row segments.
#include "mvd_common.h"

namespace mvd {

struct SweepParams {
    ViewPtrs src;        // V x (N,hs,ws,C) channel-last source features
    ViewPtrs K_src;      // V x (N,3,3)
    ViewPtrs T;          // V x (N,4,4)
    ViewOutPtrs corr;    // V x (N,S,h,w)
    ViewOutPtrs mask;    // V x (N,S,h,w)
    const float* key;    // (N,h,w,C) channel-last key features
    const float* K_key;  // (N,3,3)
    const float* invd;   // (N,S) or (1,S)
    int invd_stride;     // S if batched else 0
    int N, h, w, hs, ws, S, V;
};

constexpr int SWEEP_PX = 16;  // key pixels per workgroup

// EpipolarCoeffs.from_calib (planesweep_corr.py:262-291): 12 scalars per (view, batch element)
struct Epi {
    float a, b, c, e, f, g, h, i, j, k, l, m;
};

__device__ __forceinline__ Epi epipolar(const float* __restrict__ Kk, const float* __restrict__ Ks,
                                        const float* __restrict__ T, int h, int w, int hs, int ws) {
    const float fx = Kk[0] * (float)w, fy = Kk[4] * (float)h, cx = Kk[2] * (float)w, cy = Kk[5] * (float)h;
    const float fxo = Ks[0] * (float)ws, fyo = Ks[4] * (float)hs, cxo = Ks[2] * (float)ws, cyo = Ks[5] * (float)hs;
    const float r11 = T[0], r12 = T[1], r13 = T[2], t1 = T[3];
    const float r21 = T[4], r22 = T[5], r23 = T[6], t2 = T[7];
    const float r31 = T[8], r32 = T[9], r33 = T[10], t3 = T[11];
    Epi E;
    const float A = fxo * r11 + cxo * r31, B = fxo * r12 + cxo * r32;
    E.a = A / fx;
    E.b = B / fy;
    E.c = -(cx * A / fx) - (cy * B / fy) + (fxo * r13 + cxo * r33);
    E.e = fxo * t1 + cxo * t3;
    const float F = fyo * r21 + cyo * r31, G = fyo * r22 + cyo * r32;
    E.f = F / fx;
    E.g = G / fy;
    E.h = -(cx * F / fx) - (cy * G / fy) + (fyo * r23 + cyo * r33);
    E.i = fyo * t2 + cyo * t3;
    E.j = r31 / fx;
    E.k = r32 / fy;
    E.l = -cx * r31 / fx - cy * r32 / fy + r33;
    E.m = t3;
    return E;
}

__device__ __forceinline__ float replace_nonfinite(float v) {
    // us[isinf] = 1e9*sign(us); us[isnan] = 1e9 (planesweep_corr.py:336-338)
    if (isinf(v)) return v > 0.f ? 1e9f : -1e9f;
    if (isnan(v)) return 1e9f;
    return v;
}

template <int NJ>  // C = 64 * NJ
__global__ void __launch_bounds__(256) sweep_corr_kernel(SweepParams p) {
    extern __shared__ __attribute__((aligned(16))) float res[];  // [2][S][SWEEP_PX]
    constexpr int C = 64 * NJ;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int cl = lane & 15;  // channel lane within the group
    const int grp = lane >> 4; // plane slot within the wave
    const int y = blockIdx.y;
    const int v = blockIdx.z % p.V;
    const int n = blockIdx.z / p.V;
    const int x0 = blockIdx.x * SWEEP_PX;
    const int h = p.h, w = p.w, hs = p.hs, ws = p.ws, S = p.S;

    const Epi E = epipolar(p.K_key + n * 9, p.K_src.p[v] + n * 9, p.T.p[v] + n * 16, h, w, hs, ws);
    const float* __restrict__ src = p.src.p[v] + (size_t)n * hs * ws * C + cl * 4;
    const float* __restrict__ invd = p.invd + (size_t)n * p.invd_stride;
    const float inv_sqrt_c = 1.0f / sqrtf((float)C);
    const float fws = (float)ws, fhs = (float)hs;
    const float yc = (float)y + 0.5f;

    for (int pi = wave; pi < SWEEP_PX; pi += 4) {
        const int x = x0 + pi;
        if (x >= w) break;  // wave-uniform
        const float xc = (float)x + 0.5f;
        // u_infs_h = a*x + b*y + c etc. (planesweep_corr.py:277-290), one rounding per operation
        const float u_inf = (E.a * xc + E.b * yc) + E.c;
        const float v_inf = (E.f * xc + E.g * yc) + E.h;
        const float k_inf = (E.j * xc + E.k * yc) + E.l;
        const float z_pole = -(E.m / k_inf);  // :330

        float4 kf[NJ];
        const float* kp = p.key + (((size_t)n * h + y) * w + x) * C + cl * 4;
#pragma unroll
        for (int j = 0; j < NJ; ++j) kf[j] = *reinterpret_cast<const float4*>(kp + 64 * j);

        for (int s = grp; s < S; s += 4) {
            const float ds = invd[s];
            const float den = k_inf + E.m * ds;
            const float us = replace_nonfinite((u_inf + E.e * ds) / den);  // :334
            const float vs = replace_nonfinite((v_inf + E.i * ds) / den);  // :343
            const float zs = 1.0f / ds;                                      // :492
            const bool visible = (zs > 0.f) && (((k_inf > 0.f) && (zs > z_pole)) || ((k_inf < 0.f) && (zs < z_pole)) ||
                                               ((k_inf == 0.f) && (E.m > 0.f)));  // :499-506
            // warp(): grid = 2*u/w_x - 1 (:87-88), then grid_sample's unnormalisation
            const float ix = unnormalize_coord(2.0f * us / fws - 1.0f, fws);
            const float iy = unnormalize_coord(2.0f * vs / fhs - 1.0f, fhs);
            const Taps t = bilinear_taps(ix, iy, hs, ws);
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float* sp = src + (size_t)t.off[k] * C;
                float dot = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float4 f = *reinterpret_cast<const float4*>(sp + 64 * j);
                    dot = fmaf(kf[j].x, f.x, dot);
                    dot = fmaf(kf[j].y, f.y, dot);
                    dot = fmaf(kf[j].z, f.z, dot);
                    dot = fmaf(kf[j].w, f.w, dot);
                }
                acc = fmaf(dot, t.w[k], acc);
            }
            // sum over the 16 channel lanes of the group
            acc += __shfl_xor(acc, 8, 16);
            acc += __shfl_xor(acc, 4, 16);
            acc += __shfl_xor(acc, 2, 16);
            acc += __shfl_xor(acc, 1, 16);
            if (cl == 0) {
                // mask[mask < 0.9999] = 0; mask[mask > 0] = 1 (:101-102), times the visibility mask (:191-193)
                const float mk = (t.inb < 0.9999f || !visible) ? 0.f : 1.f;
                res[s * SWEEP_PX + pi] = acc * inv_sqrt_c * mk;
                res[(S + s) * SWEEP_PX + pi] = mk;
            }
        }
    }
    __syncthreads();
    const int npx = min(SWEEP_PX, w - x0);
    float* __restrict__ co = p.corr.p[v] + ((size_t)n * S * h + y) * w + x0;
    float* __restrict__ mo = p.mask.p[v] + ((size_t)n * S * h + y) * w + x0;
    const size_t plane = (size_t)h * w;
    for (int e = tid; e < S * SWEEP_PX; e += 256) {
        const int s = e / SWEEP_PX, px = e % SWEEP_PX;
        if (px < npx) {
            co[(size_t)s * plane + px] = res[e];
            mo[(size_t)s * plane + px] = res[S * SWEEP_PX + e];
        }
    }
}

int transpose_launch(const float* src, float* dst, int N, long long rows, long long cols, hipStream_t st);

}  // namespace mvd

extern "C" {

size_t mvd_sweep_corr_workspace_bytes(int N, int C, int h, int w, int hs, int ws, int V) {
    if (N <= 0 || C <= 0 || h <= 0 || w <= 0 || hs <= 0 || ws <= 0 || V <= 0) return 0;
    return mvd::align_up((size_t)N * C * h * w * sizeof(float), 256) +
           (size_t)V * mvd::align_up((size_t)N * C * hs * ws * sizeof(float), 256);
}

int mvd_sweep_corr_f32(const float* feat_key, const float* const* feat_src, const float* K_key,
                       const float* const* K_src, const float* const* T_src2key, const float* invdepths,
                       int invdepth_batched, int N, int C, int h, int w, int hs, int ws, int S, int V,
                       float* const* corr_out, float* const* mask_out, void* workspace, size_t workspace_bytes,
                       mvd_stream_t stream) {
    MVD_REQUIRE(feat_key && feat_src && K_key && K_src && T_src2key && invdepths && corr_out && mask_out,
                "sweep_corr: NULL argument");
    MVD_REQUIRE(N > 0 && h > 0 && w > 0 && hs > 0 && ws > 0 && S > 0, "sweep_corr: non-positive dimension");
    MVD_REQUIRE(V >= 1 && V <= MVD_MAX_VIEWS, "sweep_corr: V=%d outside 1..%d", V, MVD_MAX_VIEWS);
    MVD_REQUIRE(C % 64 == 0 && C >= 64 && C <= 512, "sweep_corr: C=%d must be a multiple of 64 in 64..512", C);
    MVD_REQUIRE(h <= 65535 && (long long)N * V <= 65535, "sweep_corr: h or N*V exceeds 65535");
    const size_t lds = (size_t)2 * S * mvd::SWEEP_PX * sizeof(float);
    MVD_REQUIRE(lds <= 160 * 1024, "sweep_corr: S=%d needs %zu B of LDS (> 160 KiB)", S, lds);
    const size_t need = mvd_sweep_corr_workspace_bytes(N, C, h, w, hs, ws, V);
    if (!workspace || workspace_bytes < need) {
        mvd::set_error("sweep_corr: workspace %zu B < required %zu B", workspace_bytes, need);
        return MVD_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* wsp = (float*)workspace;
    mvd::SweepParams p{};
    int rc = mvd::transpose_launch(feat_key, wsp, N, C, (long long)h * w, st);
    if (rc) return rc;
    p.key = wsp;
    wsp += mvd::align_up((size_t)N * C * h * w * sizeof(float), 256) / sizeof(float);
    const size_t per = mvd::align_up((size_t)N * C * hs * ws * sizeof(float), 256) / sizeof(float);
    for (int v = 0; v < V; ++v) {
        MVD_REQUIRE(feat_src[v] && K_src[v] && T_src2key[v] && corr_out[v] && mask_out[v], "sweep_corr: NULL view %d", v);
        rc = mvd::transpose_launch(feat_src[v], wsp + v * per, N, C, (long long)hs * ws, st);
        if (rc) return rc;
        p.src.p[v] = wsp + v * per;
        p.K_src.p[v] = K_src[v];
        p.T.p[v] = T_src2key[v];
        p.corr.p[v] = corr_out[v];
        p.mask.p[v] = mask_out[v];
    }
    p.K_key = K_key;
    p.invd = invdepths;
    p.invd_stride = invdepth_batched ? S : 0;
    p.N = N; p.h = h; p.w = w; p.hs = hs; p.ws = ws; p.S = S; p.V = V;
    dim3 grid((unsigned)((w + mvd::SWEEP_PX - 1) / mvd::SWEEP_PX), (unsigned)h, (unsigned)(N * V));
    mvd::timing_begin(st);
    switch (C / 64) {
#define MVD_CASE(NJ)                                                                                          \
    case NJ:                                                                                                  \
        if (lds > 64 * 1024 &&                                                                                \
            hipFuncSetAttribute((const void*)mvd::sweep_corr_kernel<NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds) != hipSuccess)                                                      \
            return mvd::launch_status("sweep_corr: LDS attribute");                                           \
        hipLaunchKernelGGL((mvd::sweep_corr_kernel<NJ>), grid, dim3(256), lds, st, p);                       \
        break;
        MVD_CASE(1) MVD_CASE(2) MVD_CASE(3) MVD_CASE(4) MVD_CASE(5) MVD_CASE(6) MVD_CASE(7) MVD_CASE(8)
#undef MVD_CASE
    }
    mvd::timing_end(st);
    return mvd::launch_status("sweep_corr");
}
}
