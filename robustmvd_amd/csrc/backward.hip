// Backward (vector-Jacobian products) of the sweep operators w.r.t. the FEATURE MAPS — SURVEY.md 8f rank 3:
//   K3  homo_warp + variance          rmvd/models/blocks/utils.py:222-268, rmvd/models/mvsnet.py:124-135
//   K1  PlanesweepCorrelation          rmvd/models/blocks/planesweep_corr.py:152-195 (grids under no_grad: :436,464,489)
//   K2  LearnedFusion view weighting   rmvd/models/blocks/learned_fusion.py:32-48
// so that the training loop (rmvd/train/multi_view_depth_training.py:231-246) can back-propagate through the engine.
// The sampling grids depend on calibration only and carry no gradient: the backward of a bilinear gather is a
// scatter-add of the incoming gradient times the same weights, done with no-return float atomics
// (global_atomic_add_f32; MI355X_MICROARCH.md "Global float atomics").  These kernels favour simplicity over speed: the
// forward kernels are the hot path, training runs at reduced sizes.  Summation order differs from run to run (atomics),
// results agree with autograd through the reference to ~1e-5 relative (tests/golden/g10_grads.npz).
#include "mvd_common.h"

namespace mvd {

__device__ __forceinline__ void atomic_add4(float* p, float4 v) {
    unsafeAtomicAdd(p + 0, v.x); unsafeAtomicAdd(p + 1, v.y); unsafeAtomicAdd(p + 2, v.z); unsafeAtomicAdd(p + 3, v.w);
}

// ---------------------------------------------------------------------------------------------------------------
// K3 backward.  Thread = (pixel, channel quad); loops over the D planes in chunks of 8.  Per chunk: pass 1 gathers every view's
// samples to form the planes' means, pass 2 gathers again per view (no per-view register array) and scatters
// 2 g (x_v - mean) / (V+1) times the bilinear weights into that view's gradient map, runs of planes that hit the same source
// cell summed in registers first.  Sampling positions: the folded form of the forward kernel.
struct WarpBwdParams {
    ViewPtrs src;            // V x (B,h+3,w+3,C) zero-bordered channel-last source features
    ViewOutPtrs gsrc;        // V x (B,h+3,w+3,C) zero-initialised gradient maps (border entries receive the padding's share)
    const float* key;        // (B,h+3,w+3,C)
    float* gkey;             // (B,h+3,w+3,C), interior written (not accumulated)
    const float* M;          // (V,B,12) composed transforms
    const float* depth;      // (B,D)
    const float* gvar;       // (B,D,h,w,C) channel-last
    int B, C, D, h, w, V;
};

__global__ void __launch_bounds__(256) warp_variance_backward_kernel(WarpBwdParams p) {
    const int lpp = p.C / 4;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long npix = (long long)p.B * p.h * p.w;
    if (t >= npix * lpp) return;
    const int q = (int)(t % lpp);
    long long pix = t / lpp;
    const int x = (int)(pix % p.w); pix /= p.w;
    const int y = (int)(pix % p.h);
    const int b = (int)(pix / p.h);
    const int h = p.h, w = p.w, C = p.C, D = p.D, V = p.V;
    const int W2 = w + 3;
    const size_t img = (size_t)(h + 3) * W2 * C;
    const float sx = (float)w / (float)(w - 1), sy = (float)h / (float)(h - 1);
    const float fx = (float)x, fy = (float)y, xhi = (float)w, yhi = (float)h;
    const float inv_nv = 1.0f / (float)(V + 1);
    const size_t self = ((size_t)(y + 1) * W2 + (x + 1)) * C + q * 4;
    const float4 k = *reinterpret_cast<const float4*>(p.key + b * img + self);

    struct Loc { size_t o; float w00, w10, w01, w11; };
    auto locate = [&](int v, float depth) {
        const float* __restrict__ M = p.M + ((size_t)v * p.B + b) * 12;
        const float ax = fmaf(M[0], fx, fmaf(M[1], fy, M[2])), ay = fmaf(M[4], fx, fmaf(M[5], fy, M[6]));
        const float az = fmaf(M[8], fx, fmaf(M[9], fy, M[10]));
        const float X = fmaf(ax, depth, M[3]), Y = fmaf(ay, depth, M[7]), Z = fmaf(az, depth, M[11]);
        const float rz = __builtin_amdgcn_rcpf(Z);
        float ix = fmaf(X * rz, sx, -0.5f), iy = fmaf(Y * rz, sy, -0.5f);
        ix = __builtin_amdgcn_fmed3f(ix, -1.0f, xhi);
        iy = __builtin_amdgcn_fmed3f(iy, -1.0f, yhi);
        const float xf = floorf(ix), yf = floorf(iy);
        const float wx = ix - xf, wy = iy - yf, ux = 1.0f - wx, uy = 1.0f - wy;
        Loc L;
        L.o = ((size_t)((int)yf + 1) * W2 + ((int)xf + 1)) * C + q * 4;
        L.w00 = ux * uy; L.w10 = wx * uy; L.w01 = ux * wy; L.w11 = wx * wy;
        return L;
    };
    auto sample = [&](const float* __restrict__ f, const Loc& L) {
        const float4 a = *reinterpret_cast<const float4*>(f + L.o), bq = *reinterpret_cast<const float4*>(f + L.o + C);
        const float4 c = *reinterpret_cast<const float4*>(f + L.o + (size_t)W2 * C), d = *reinterpret_cast<const float4*>(f + L.o + (size_t)W2 * C + C);
        return make_float4(fmaf(d.x, L.w11, fmaf(c.x, L.w01, fmaf(bq.x, L.w10, a.x * L.w00))),
                           fmaf(d.y, L.w11, fmaf(c.y, L.w01, fmaf(bq.y, L.w10, a.y * L.w00))),
                           fmaf(d.z, L.w11, fmaf(c.z, L.w01, fmaf(bq.z, L.w10, a.z * L.w00))),
                           fmaf(d.w, L.w11, fmaf(c.w, L.w01, fmaf(bq.w, L.w10, a.w * L.w00))));
    };

    // Planes in chunks of BWD_DZ: the chunk's means stay in registers, then each view walks the chunk with ONE pending 2 x 2 cell of
    // tap gradients in registers: consecutive planes of a pixel mostly sample the same source cell (the forward kernel's tap
    // reuse), so the four taps' shares are summed in registers and go out as atomics only when the cell changes (and at the end
    // of the chunk) — about 2.5x fewer atomics at the headline poses.
    constexpr int DZ = 8;
    const float c2 = 2.0f * inv_nv;
    float4 gk = make_float4(0, 0, 0, 0);
    for (int d0 = 0; d0 < D; d0 += DZ) {
        float4 mean[DZ], gs[DZ];  // gs = g * 2 / (V + 1)
#pragma unroll
        for (int dd = 0; dd < DZ; ++dd) {
            mean[dd] = gs[dd] = make_float4(0, 0, 0, 0);
            const int d = d0 + dd;
            if (d >= D) continue;
            const float depth = p.depth[(size_t)b * D + d];
            const float4 g = *reinterpret_cast<const float4*>(p.gvar + ((((size_t)b * D + d) * h + y) * w + x) * C + q * 4);
            float4 sum = k;
            for (int v = 0; v < V; ++v) {
                const float4 xv = sample(p.src.p[v] + b * img, locate(v, depth));
                sum.x += xv.x; sum.y += xv.y; sum.z += xv.z; sum.w += xv.w;
            }
            mean[dd] = make_float4(sum.x * inv_nv, sum.y * inv_nv, sum.z * inv_nv, sum.w * inv_nv);
            gs[dd] = make_float4(g.x * c2, g.y * c2, g.z * c2, g.w * c2);
            gk.x += gs[dd].x * (k.x - mean[dd].x); gk.y += gs[dd].y * (k.y - mean[dd].y);
            gk.z += gs[dd].z * (k.z - mean[dd].z); gk.w += gs[dd].w * (k.w - mean[dd].w);
        }
        for (int v = 0; v < V; ++v) {
            const float* __restrict__ f = p.src.p[v] + b * img;
            float* __restrict__ gv = p.gsrc.p[v] + b * img;
            constexpr size_t NONE = ~(size_t)0;
            size_t pend = NONE;
            float4 t00 = make_float4(0, 0, 0, 0), t10 = t00, t01 = t00, t11 = t00;
            auto flush = [&]() {
                float* go = gv + pend;
                atomic_add4(go, t00);
                atomic_add4(go + C, t10);
                atomic_add4(go + (size_t)W2 * C, t01);
                atomic_add4(go + (size_t)W2 * C + C, t11);
            };
#pragma unroll
            for (int dd = 0; dd < DZ; ++dd) {
                const int d = d0 + dd;
                if (d >= D) continue;
                const Loc L = locate(v, p.depth[(size_t)b * D + d]);
                const float4 xv = sample(f, L);
                const float4 gx = make_float4(gs[dd].x * (xv.x - mean[dd].x), gs[dd].y * (xv.y - mean[dd].y),
                                              gs[dd].z * (xv.z - mean[dd].z), gs[dd].w * (xv.w - mean[dd].w));
                if (L.o != pend) {
                    if (pend != NONE) flush();
                    pend = L.o;
                    t00 = make_float4(gx.x * L.w00, gx.y * L.w00, gx.z * L.w00, gx.w * L.w00);
                    t10 = make_float4(gx.x * L.w10, gx.y * L.w10, gx.z * L.w10, gx.w * L.w10);
                    t01 = make_float4(gx.x * L.w01, gx.y * L.w01, gx.z * L.w01, gx.w * L.w01);
                    t11 = make_float4(gx.x * L.w11, gx.y * L.w11, gx.z * L.w11, gx.w * L.w11);
                } else {
                    t00.x = fmaf(gx.x, L.w00, t00.x); t00.y = fmaf(gx.y, L.w00, t00.y); t00.z = fmaf(gx.z, L.w00, t00.z); t00.w = fmaf(gx.w, L.w00, t00.w);
                    t10.x = fmaf(gx.x, L.w10, t10.x); t10.y = fmaf(gx.y, L.w10, t10.y); t10.z = fmaf(gx.z, L.w10, t10.z); t10.w = fmaf(gx.w, L.w10, t10.w);
                    t01.x = fmaf(gx.x, L.w01, t01.x); t01.y = fmaf(gx.y, L.w01, t01.y); t01.z = fmaf(gx.z, L.w01, t01.z); t01.w = fmaf(gx.w, L.w01, t01.w);
                    t11.x = fmaf(gx.x, L.w11, t11.x); t11.y = fmaf(gx.y, L.w11, t11.y); t11.z = fmaf(gx.z, L.w11, t11.z); t11.w = fmaf(gx.w, L.w11, t11.w);
                }
            }
            if (pend != NONE) flush();
        }
    }
    *reinterpret_cast<float4*>(p.gkey + b * img + self) = gk;
}

__global__ void compose_transforms_bwd_kernel(ViewPtrs proj, const float* __restrict__ key_proj_inv, int B, int V, float* __restrict__ M) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= V * B * 12) return;
    const int j = e % 4, i = (e / 4) % 3, b = (e / 12) % B, v = e / (12 * B);
    const float* P = proj.p[v] + b * 16;
    const float* Q = key_proj_inv + b * 16;
    float acc = P[i * 4 + 0] * Q[0 * 4 + j];  // the same fmaf chain as the forward's compose_transforms_kernel
    acc = fmaf(P[i * 4 + 1], Q[1 * 4 + j], acc);
    acc = fmaf(P[i * 4 + 2], Q[2 * 4 + j], acc);
    acc = fmaf(P[i * 4 + 3], Q[3 * 4 + j], acc);
    M[e] = acc;
}

// ---------------------------------------------------------------------------------------------------------------
// K1 backward.  One wave per key pixel, lane = channel slice (C = 64 NJ), loops over views and planes; geometry (the
// forward's own operation chain: it decides the 0/1 mask) is evaluated by lane = plane in passes of 64 and broadcast
// with readlane.  Consecutive planes that share a 2x2 source cell accumulate their tap gradients in registers and
// flush them with atomics when the cell changes (~4x fewer atomics).
struct Epi { float a, b, c, e, f, g, h, i, j, k, l, m; };
__device__ __forceinline__ Epi epipolar_b(const float* __restrict__ Kk, const float* __restrict__ Ks, const float* __restrict__ T,
                                          int h, int w, int hs, int ws) {
    const float fx = Kk[0] * (float)w, fy = Kk[4] * (float)h, cx = Kk[2] * (float)w, cy = Kk[5] * (float)h;
    const float fxo = Ks[0] * (float)ws, fyo = Ks[4] * (float)hs, cxo = Ks[2] * (float)ws, cyo = Ks[5] * (float)hs;
    const float r11 = T[0], r12 = T[1], r13 = T[2], t1 = T[3], r21 = T[4], r22 = T[5], r23 = T[6], t2 = T[7];
    const float r31 = T[8], r32 = T[9], r33 = T[10], t3 = T[11];
    Epi E;
    const float A = fxo * r11 + cxo * r31, Bq = fxo * r12 + cxo * r32;
    E.a = A / fx; E.b = Bq / fy;
    E.c = -(cx * A / fx) - (cy * Bq / fy) + (fxo * r13 + cxo * r33);
    E.e = fxo * t1 + cxo * t3;
    const float F = fyo * r21 + cyo * r31, G = fyo * r22 + cyo * r32;
    E.f = F / fx; E.g = G / fy;
    E.h = -(cx * F / fx) - (cy * G / fy) + (fyo * r23 + cyo * r33);
    E.i = fyo * t2 + cyo * t3;
    E.j = r31 / fx; E.k = r32 / fy;
    E.l = -cx * r31 / fx - cy * r32 / fy + r33;
    E.m = t3;
    return E;
}
__device__ __forceinline__ float fix_nonfinite(float v) {
    if (isinf(v)) return v > 0.f ? 1e9f : -1e9f;
    if (isnan(v)) return 1e9f;
    return v;
}

struct SweepBwdParams {
    ViewPtrs src;        // V x (N,hs+3,ws+3,C) zero-bordered channel-last source features
    ViewPtrs K_src, T;   // V x (N,3,3), V x (N,4,4)
    ViewPtrs gcorr;      // V x (N,S,h,w)
    ViewOutPtrs gsrc;    // V x (N,hs+3,ws+3,C), zero-initialised
    const float* key;    // (N,h,w,C) channel-last
    float* gkey;         // (N,h,w,C)
    const float* K_key;  // (N,3,3)
    const float* invd;
    int invd_stride;
    int invd_per_pixel;
    float corr_scale;
    int N, h, w, hs, ws, S, V;
};

template <int NJ>
__global__ void __launch_bounds__(256) sweep_corr_backward_kernel(SweepBwdParams p) {
    constexpr int C = 64 * NJ;
    const int lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (long long)p.N * p.h * p.w) return;  // wave-uniform
    const int x = (int)(wid % p.w), y = (int)((wid / p.w) % p.h), n = (int)(wid / ((long long)p.w * p.h));
    const int h = p.h, w = p.w, hs = p.hs, ws = p.ws, S = p.S, W2 = ws + 3;
    const float inv_sqrt_c = p.corr_scale;
    const float fws = (float)ws, fhs = (float)hs, xc = (float)x + 0.5f, yc = (float)y + 0.5f;
    const float* __restrict__ invd = p.invd + (size_t)n * p.invd_stride;
    float kf[NJ], gk[NJ];
    const size_t koff = (((size_t)n * h + y) * w + x) * C + lane * NJ;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { kf[j] = p.key[koff + j]; gk[j] = 0.f; }

    for (int v = 0; v < p.V; ++v) {
        const Epi E = epipolar_b(p.K_key + n * 9, p.K_src.p[v] + n * 9, p.T.p[v] + n * 16, h, w, hs, ws);
        const float u_inf = (E.a * xc + E.b * yc) + E.c, v_inf = (E.f * xc + E.g * yc) + E.h, k_inf = (E.j * xc + E.k * yc) + E.l;
        const float z_pole = -(E.m / k_inf);
        const size_t simg = (size_t)n * (hs + 3) * W2 * C;
        const float* __restrict__ src = p.src.p[v] + simg + lane * NJ;
        float* __restrict__ gsrc = p.gsrc.p[v] + simg + lane * NJ;
        const float* __restrict__ gc = p.gcorr.p[v] + (((size_t)n * S) * h + y) * w + x;
        int cur = -1;
        float tap[4][NJ], gt[4][NJ];
        auto flush = [&]() {
            if (cur < 0) return;
            float* g0 = gsrc + (size_t)cur * C;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                unsafeAtomicAdd(g0 + j, gt[0][j]); unsafeAtomicAdd(g0 + C + j, gt[1][j]);
                unsafeAtomicAdd(g0 + (size_t)W2 * C + j, gt[2][j]); unsafeAtomicAdd(g0 + (size_t)W2 * C + C + j, gt[3][j]);
            }
        };
        for (int s0 = 0; s0 < S; s0 += 64) {
            const int s = s0 + lane;
            const bool live = s < S;
            const int sc = live ? s : S - 1;
            const float ds = p.invd_per_pixel ? p.invd[(((size_t)n * S + sc) * h + y) * w + x] : invd[sc];
            const float den = k_inf + E.m * ds;
            const float us = fix_nonfinite((u_inf + E.e * ds) / den), vs = fix_nonfinite((v_inf + E.i * ds) / den);
            const float zs = 1.0f / ds;
            const bool visible = (zs > 0.f) && (((k_inf > 0.f) && (zs > z_pole)) || ((k_inf < 0.f) && (zs < z_pole)) ||
                                               ((k_inf == 0.f) && (E.m > 0.f)));
            const float ix = unnormalize_coord(2.0f * us / fws - 1.0f, fws), iy = unnormalize_coord(2.0f * vs / fhs - 1.0f, fhs);
            const Taps t = bilinear_taps(ix, iy, hs, ws);
            const float mk = (t.inb < 0.9999f || !visible) ? 0.f : 1.f;
            const int cx = (int)fminf(fmaxf(floorf(ix), -1.0f), (float)(ws - 1)) + 1;
            const int cy = (int)fminf(fmaxf(floorf(iy), -1.0f), (float)(hs - 1)) + 1;
            const int cell = cy * W2 + cx;
            const float gcoef = live ? gc[(size_t)s * h * w] * mk * inv_sqrt_c : 0.f;
            const int npl = min(64, S - s0);
            for (int i = 0; i < npl; ++i) {  // wave-uniform
                const float gci = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gcoef), i));
                if (gci == 0.f) continue;
                const int ci = __builtin_amdgcn_readlane(cell, i);
                if (ci != cur) {
                    flush();
                    cur = ci;
                    const float* s00 = src + (size_t)ci * C;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        tap[0][j] = s00[j]; tap[1][j] = s00[C + j];
                        tap[2][j] = s00[(size_t)W2 * C + j]; tap[3][j] = s00[(size_t)W2 * C + C + j];
                        gt[0][j] = gt[1][j] = gt[2][j] = gt[3][j] = 0.f;
                    }
                }
                float wk[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) wk[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t.w[k]), i));
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float samp = fmaf(tap[3][j], wk[3], fmaf(tap[2][j], wk[2], fmaf(tap[1][j], wk[1], tap[0][j] * wk[0])));
                    gk[j] = fmaf(gci, samp, gk[j]);
                    const float gkf = gci * kf[j];
#pragma unroll
                    for (int k = 0; k < 4; ++k) gt[k][j] = fmaf(gkf, wk[k], gt[k][j]);
                }
            }
        }
        flush();
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) p.gkey[koff + j] = gk[j];
}

// ---------------------------------------------------------------------------------------------------------------
// K2 backward.  One wave per (n, y, x): lanes stride over the S planes; the gradient w.r.t. the per-view score (one value
// per pixel) is the wave-reduced sum over planes pushed through the softmax Jacobian.
struct FuseBwdParams {
    ViewPtrs corr, mask, score, dummy;
    ViewOutPtrs gcorr, gscore;
    const float* gfused;
    int N, S, h, w, V;
};

__global__ void __launch_bounds__(256) fuse_views_backward_kernel(FuseBwdParams p) {
    const int lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long hw = (long long)p.h * p.w;
    if (wid >= p.N * hw) return;
    const int n = (int)(wid / hw);
    const long long pix = wid % hw;
    const int V = p.V, S = p.S;
    float pr[MVD_MAX_VIEWS];
    float mx = -3.4e38f;
    for (int v = 0; v < V; ++v) { pr[v] = p.score.p[v][n * hw + pix]; mx = fmaxf(mx, pr[v]); }
    float den = 0.f;
    for (int v = 0; v < V; ++v) { pr[v] = expf(pr[v] - mx); den += pr[v]; }
    for (int v = 0; v < V; ++v) pr[v] /= den;
    float dp[MVD_MAX_VIEWS];
    for (int v = 0; v < V; ++v) dp[v] = 0.f;
    for (int s = lane; s < S; s += 64) {
        const size_t o = ((size_t)n * S + s) * hw + pix;
        float W = 0.f, num = 0.f;
        for (int v = 0; v < V; ++v) {
            const float u = (pr[v] + 1e-9f) * p.mask.p[v][o];
            W += u;
            num = fmaf(p.corr.p[v][o], u, num);
        }
        const float g = (W != 0.f) ? p.gfused[o] : 0.f;
        const float d = W + 1e-9f, q = num / d;
        for (int v = 0; v < V; ++v) {
            const float m = p.mask.p[v][o], u = (pr[v] + 1e-9f) * m;
            p.gcorr.p[v][o] = g * u / d;
            dp[v] = fmaf(g * (p.corr.p[v][o] - q) / d, m, dp[v]);
        }
    }
    float dot = 0.f;
    for (int v = 0; v < V; ++v) {
        float t = dp[v];
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
        dp[v] = t;
        dot = fmaf(pr[v], t, dot);
    }
    if (lane == 0)
        for (int v = 0; v < V; ++v) p.gscore.p[v][n * hw + pix] = pr[v] * (dp[v] - dot);
}

}  // namespace mvd

extern "C" {

size_t mvd_warp_variance_backward_workspace_bytes(int B) {
    return B > 0 ? mvd::align_up((size_t)MVD_MAX_VIEWS * B * 12 * sizeof(float), 256) : 0;
}

int mvd_warp_variance_backward_f32(const float* key_feat, const float* const* src_feat, const float* const* src_proj,
                                   const float* key_proj_inv, const float* depth_values, const float* grad_var, int B, int C,
                                   int D, int h, int w, int V, float* grad_key, float* const* grad_src, void* workspace,
                                   size_t workspace_bytes, mvd_stream_t stream) {
    using namespace mvd;
    MVD_REQUIRE(key_feat && src_feat && src_proj && key_proj_inv && depth_values && grad_var && grad_key && grad_src,
                "warp_variance_backward: NULL argument");
    MVD_REQUIRE(B > 0 && D > 0 && h > 1 && w > 1 && V >= 1 && V <= MVD_MAX_VIEWS, "warp_variance_backward: bad dimensions");
    MVD_REQUIRE(C >= 4 && C % 4 == 0, "warp_variance_backward: C=%d must be a positive multiple of 4", C);
    const size_t need = mvd_warp_variance_backward_workspace_bytes(B);
    if (!workspace || workspace_bytes < need) {
        set_error("warp_variance_backward: workspace %zu B < required %zu B", workspace_bytes, need);
        return MVD_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    WarpBwdParams p{};
    ViewPtrs proj{};
    const size_t slot = (size_t)B * (h + 3) * (w + 3) * C * sizeof(float);
    for (int v = 0; v < V; ++v) {
        MVD_REQUIRE(src_feat[v] && src_proj[v] && grad_src[v], "warp_variance_backward: NULL view %d", v);
        p.src.p[v] = src_feat[v];
        p.gsrc.p[v] = grad_src[v];
        proj.p[v] = src_proj[v];
        if (hipMemsetAsync(grad_src[v], 0, slot, st) != hipSuccess) return launch_status("warp_variance_backward: memset");
    }
    if (hipMemsetAsync(grad_key, 0, slot, st) != hipSuccess) return launch_status("warp_variance_backward: memset");
    p.M = (float*)workspace;
    hipLaunchKernelGGL(compose_transforms_bwd_kernel, dim3((unsigned)((V * B * 12 + 255) / 256)), dim3(256), 0, st, proj,
                       key_proj_inv, B, V, (float*)workspace);
    p.key = key_feat; p.gkey = grad_key; p.depth = depth_values; p.gvar = grad_var;
    p.B = B; p.C = C; p.D = D; p.h = h; p.w = w; p.V = V;
    const long long nthr = (long long)B * h * w * (C / 4), nblk = (nthr + 255) / 256;
    MVD_REQUIRE(nblk <= 0x7fffffffLL, "warp_variance_backward: grid too large");
    hipLaunchKernelGGL(warp_variance_backward_kernel, dim3((unsigned)nblk), dim3(256), 0, st, p);
    return launch_status("warp_variance_backward");
}

int mvd_sweep_corr_backward_f32(const float* feat_key, const float* const* feat_src, const float* K_key,
                                const float* const* K_src, const float* const* T_src2key, const float* invdepths,
                                int invdepth_mode, float corr_scale, const float* const* grad_corr, int N, int C, int h, int w,
                                int hs, int ws, int S, int V, float* grad_key, float* const* grad_src, mvd_stream_t stream) {
    using namespace mvd;
    MVD_REQUIRE(invdepth_mode >= MVD_INVDEPTH_SHARED && invdepth_mode <= MVD_INVDEPTH_PER_PIXEL, "sweep_corr_backward: invdepth_mode %d", invdepth_mode);
    MVD_REQUIRE(feat_key && feat_src && K_key && K_src && T_src2key && invdepths && grad_corr && grad_key && grad_src,
                "sweep_corr_backward: NULL argument");
    MVD_REQUIRE(N > 0 && h > 0 && w > 0 && hs > 0 && ws > 0 && S > 0 && V >= 1 && V <= MVD_MAX_VIEWS, "sweep_corr_backward: bad dimensions");
    MVD_REQUIRE(C == 64 || C == 128 || C == 192 || C == 256, "sweep_corr_backward: C=%d unsupported (64, 128, 192, 256)", C);
    hipStream_t st = (hipStream_t)stream;
    SweepBwdParams p{};
    const size_t slot = (size_t)N * (hs + 3) * (ws + 3) * C * sizeof(float);
    for (int v = 0; v < V; ++v) {
        MVD_REQUIRE(feat_src[v] && K_src[v] && T_src2key[v] && grad_corr[v] && grad_src[v], "sweep_corr_backward: NULL view %d", v);
        p.src.p[v] = feat_src[v]; p.K_src.p[v] = K_src[v]; p.T.p[v] = T_src2key[v]; p.gcorr.p[v] = grad_corr[v];
        p.gsrc.p[v] = grad_src[v];
        if (hipMemsetAsync(grad_src[v], 0, slot, st) != hipSuccess) return launch_status("sweep_corr_backward: memset");
    }
    p.key = feat_key; p.gkey = grad_key; p.K_key = K_key; p.invd = invdepths;
    p.invd_stride = invdepth_mode == MVD_INVDEPTH_BATCHED ? S : 0;
    p.invd_per_pixel = invdepth_mode == MVD_INVDEPTH_PER_PIXEL;
    p.corr_scale = corr_scale;
    p.N = N; p.h = h; p.w = w; p.hs = hs; p.ws = ws; p.S = S; p.V = V;
    const long long nwave = (long long)N * h * w, nblk = (nwave + 3) / 4;
    MVD_REQUIRE(nblk <= 0x7fffffffLL, "sweep_corr_backward: grid too large");
    switch (C / 64) {
        case 1: hipLaunchKernelGGL(sweep_corr_backward_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL(sweep_corr_backward_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL(sweep_corr_backward_kernel<3>, dim3((unsigned)nblk), dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL(sweep_corr_backward_kernel<4>, dim3((unsigned)nblk), dim3(256), 0, st, p); break;
    }
    return launch_status("sweep_corr_backward");
}

int mvd_fuse_views_backward_f32(const float* const* corr, const float* const* mask, const float* const* score,
                                const float* grad_fused, int N, int S, int h, int w, int V, float* const* grad_corr,
                                float* const* grad_score, mvd_stream_t stream) {
    using namespace mvd;
    MVD_REQUIRE(corr && mask && score && grad_fused && grad_corr && grad_score, "fuse_views_backward: NULL argument");
    MVD_REQUIRE(N > 0 && S > 0 && h > 0 && w > 0 && V >= 2 && V <= MVD_MAX_VIEWS, "fuse_views_backward: bad dimensions (V >= 2)");
    FuseBwdParams p{};
    for (int v = 0; v < V; ++v) {
        MVD_REQUIRE(corr[v] && mask[v] && score[v] && grad_corr[v] && grad_score[v], "fuse_views_backward: NULL view %d", v);
        p.corr.p[v] = corr[v]; p.mask.p[v] = mask[v]; p.score.p[v] = score[v];
        p.gcorr.p[v] = grad_corr[v]; p.gscore.p[v] = grad_score[v];
    }
    p.gfused = grad_fused; p.N = N; p.S = S; p.h = h; p.w = w; p.V = V;
    const long long nwave = (long long)N * h * w, nblk = (nwave + 3) / 4;
    MVD_REQUIRE(nblk <= 0x7fffffffLL, "fuse_views_backward: grid too large");
    hipLaunchKernelGGL(fuse_views_backward_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status("fuse_views_backward");
}
}
