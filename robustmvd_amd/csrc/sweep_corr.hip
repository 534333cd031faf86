// K1 — inverse-depth plane sweep with dot-product correlation (Path A, robust_mvd).
// Replaces PlanesweepCorrelation.forward (rmvd/models/blocks/planesweep_corr.py:396-521):
// epipolar coefficients (:228-300), sampling grids with non-finite replacement (:333-349), visibility
// mask (:489-512) and TorchCorr (:152-195, warp() :49-104) for all V source views in one launch.
//
// The reference materialises the (h*w)x(hs*ws) all-pairs matrix per view (764 MB at 96x144) and interpolates it.
// Here the same quantity is evaluated LAZILY along each key pixel's epipolar segment: consecutive inverse-depth
// planes move the sample by 0.07-0.3 px, so most planes share the 2x2 source cell of their predecessor, and the
// four dot products <f_key, f_src(tap)> depend only on the cell.  One wave owns one (key pixel, view):
//   * geometry: lane l evaluates plane s0 + l (64 planes per pass, no redundancy): sample position, bilinear
//     weights, in-bounds mask, visibility, and the index of its 2x2 cell in the zero-bordered source copy;
//   * cells: a ballot over "my cell differs from my neighbour's" lists the distinct cells of the pass; for each one
//     (wave-uniform loop) the 64 lanes switch role to channel slices, load the four taps (1 KB each at C = 256:
//     one fully coalesced instruction per tap), form the four dots with a 7-shuffle multi-value reduction, and
//     every plane lane whose cell it is blends them with its own weights.
// Neither the all-pairs matrix nor the sampling grids ever exist in memory, and the gather + reduction work is
// per distinct cell (~60 per pixel and view at 96x144x256 planes) instead of per plane (256).
// HBM traffic is the feature maps in and 2*V*S*h*w floats out; the kernel is bound by L1 gather bandwidth and
// vector ALU, not HBM.  Results are staged in LDS and written as 64-B row segments.
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
    const float* invd;   // (1,S), (N,S) or per key pixel (N,S,h,w)
    int invd_stride;     // S if batched else 0 (ignored when per pixel)
    int invd_per_pixel;  // planesweep_corr.py:465-487 accepts (N,S), (N,S,H) and (N,S,H,W) sampling inverse depths
    float corr_scale;    // 1/sqrt(C) for normalize="dim" (planesweep_corr.py:186), 1 otherwise
    int N, h, w, hs, ws, S, V;
    int out_ps;          // 0: outputs (N,S,h,w); > 0: pixel-major (N,h,w,S) with pixels out_ps floats apart (mvd_sweep_corr_nhwc_f32)
    float* amax;         // optional: max |corr| over all views' outputs, raised atomically (pixel-major entry point; caller zeroes)
    int tiles_x, per_xcd, total;  // 1-D grid: workgroup b works on unit (b % 8) * per_xcd + b / 8 of (view, batch, row, x tile)
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

// sum of 4 values over the 64 lanes with 7 shuffles: after two select+exchange steps lane l holds a partial of value
// (l & 3), four butterfly steps finish it; value k is read back from lane k.
__device__ __forceinline__ void wave_sum4(float& v0, float& v1, float& v2, float& v3, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    float a = b0 ? v1 : v0, x = b0 ? v0 : v1;
    a += __shfl_xor(x, 1);
    float c = b0 ? v3 : v2, y = b0 ? v2 : v3;
    c += __shfl_xor(y, 1);
    float e = b1 ? c : a, z = b1 ? a : c;
    e += __shfl_xor(z, 2);
    e += __shfl_xor(e, 4);
    e += __shfl_xor(e, 8);
    e += __shfl_xor(e, 16);
    e += __shfl_xor(e, 32);
    v0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), 0));
    v1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), 1));
    v2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), 2));
    v3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), 3));
}

// the same for 8 values with 10 shuffles (two cells per pass): three select+exchange steps leave a partial of value
// (l & 7) in lane l, three butterfly steps finish it.  Every value is summed over the lanes in the same xor-1, 2, 4, 8,
// 16, 32 order as in wave_sum4, so the results are bit-identical to two wave_sum4 calls.
__device__ __forceinline__ void wave_sum8(float (&v)[8], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    float s1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float keep = b0 ? v[2 * k + 1] : v[2 * k], give = b0 ? v[2 * k] : v[2 * k + 1];
        s1[k] = keep + __shfl_xor(give, 1);
    }
    float s2[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float keep = b1 ? s1[2 * k + 1] : s1[2 * k], give = b1 ? s1[2 * k] : s1[2 * k + 1];
        s2[k] = keep + __shfl_xor(give, 2);
    }
    const float keep = b2 ? s2[1] : s2[0], give = b2 ? s2[0] : s2[1];
    float e = keep + __shfl_xor(give, 4);
    e += __shfl_xor(e, 8);
    e += __shfl_xor(e, 16);
    e += __shfl_xor(e, 32);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), k));
}

template <int NJ>  // C = 64 * NJ: lane l owns channels [l*NJ, l*NJ + NJ)
__global__ void __launch_bounds__(256) sweep_corr_kernel(SweepParams p) {
    extern __shared__ __attribute__((aligned(16))) float res[];  // [2][S][SWEEP_PX]
    constexpr int C = 64 * NJ;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // Units in launch order.  (Experiments library, MVD_K1_XCD_ORDER: XCD k takes the k-th contiguous eighth of the units, so that an
    // XCD's L2 sees neighbouring rows of one view only.  Measured 15 % SLOWER, 0.56 -> 0.65 ms: in launch order all XCDs sweep the
    // same band of the source maps at the same time and share it through the Infinity Cache.)
    int u = (int)(blockIdx.x % 8) * p.per_xcd + (int)(blockIdx.x / 8);
    if (p.per_xcd == 0) u = blockIdx.x;
    if (u >= p.total) return;
    const int xt = u % p.tiles_x; u /= p.tiles_x;
    const int y = u % p.h; u /= p.h;
    const int v = u % p.V;
    const int n = u / p.V;
    const int x0 = xt * SWEEP_PX;
    const int h = p.h, w = p.w, hs = p.hs, ws = p.ws, S = p.S;
    const int W2 = ws + 3;  // zero-bordered source: pixel (yy, xx) at padded (yy+1, xx+1)

    const Epi E = epipolar(p.K_key + n * 9, p.K_src.p[v] + n * 9, p.T.p[v] + n * 16, h, w, hs, ws);
    const float* __restrict__ src = p.src.p[v] + (size_t)n * (hs + 3) * W2 * C + lane * NJ;
    const float* __restrict__ invd = p.invd + (size_t)n * p.invd_stride;
    const float inv_sqrt_c = p.corr_scale;
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

        float kf[NJ];
        const float* kp = p.key + (((size_t)n * h + y) * w + x) * C + lane * NJ;
#pragma unroll
        for (int j = 0; j < NJ; ++j) kf[j] = kp[j];

        for (int s0 = 0; s0 < S; s0 += 64) {
            // ---- geometry of plane s0 + lane ----
            const int s = s0 + lane;
            const bool live = s < S;
            const int sc = live ? s : S - 1;
            const float ds = p.invd_per_pixel ? p.invd[(((size_t)n * S + sc) * h + y) * w + x] : invd[sc];
            const float den = k_inf + E.m * ds;
            const float us = replace_nonfinite((u_inf + E.e * ds) / den);  // :334
            const float vs = replace_nonfinite((v_inf + E.i * ds) / den);  // :343
            const float zs = 1.0f / ds;                                      // :492
            const bool visible = (zs > 0.f) && (((k_inf > 0.f) && (zs > z_pole)) || ((k_inf < 0.f) && (zs < z_pole)) ||
                                               ((k_inf == 0.f) && (E.m > 0.f)));  // :499-506
            // warp(): grid = 2*u/w_x - 1 (:87-88), then grid_sample's unnormalisation
            const float ix = unnormalize_coord(2.0f * us / fws - 1.0f, fws);
            const float iy = unnormalize_coord(2.0f * vs / fhs - 1.0f, fhs);
            const Taps t = bilinear_taps(ix, iy, hs, ws);  // weights are 0 on out-of-image taps
            // mask[mask < 0.9999] = 0; mask[mask > 0] = 1 (:101-102), times the visibility mask (:191-193)
            const float mk = (t.inb < 0.9999f || !visible) ? 0.f : 1.f;
            // 2x2 cell in the zero-bordered copy; a cell entirely outside the image has all-zero weights, so which
            // (valid) cell stands in for it does not matter
            const int cx = (int)fminf(fmaxf(floorf(ix), -1.0f), (float)(ws - 1)) + 1;
            const int cy = (int)fminf(fmaxf(floorf(iy), -1.0f), (float)(hs - 1)) + 1;
            const int cell = live ? cy * W2 + cx : -1;

            // ---- distinct cells of this pass ----
            const int prev = __shfl_up(cell, 1);
            unsigned long long todo = __ballot(live && (lane == 0 || cell != prev));
            float acc = 0.f;
            // two cells per step: their 8 tap loads are in flight together and one 10-shuffle reduction serves both (four per
            // step measured no better).  The loop is serial in `todo`, so the taps of step i+1 are fetched (into a second
            // register set) BEFORE step i is reduced: without that every step exposed a full L2 latency (round 1: 0.59 ms).
            float ta[8][NJ], tb[8][NJ];
            int ca0 = 0, cb0 = 0, ca1 = 0, cb1 = 0;
            auto pick = [&](int& ca, int& cb) {  // wave-uniform
                if (!todo) return false;
                ca = __builtin_amdgcn_readlane(cell, __builtin_ctzll(todo));
                todo &= todo - 1;
                cb = ca;  // odd count: the last step does the same cell twice
                if (todo) {
                    cb = __builtin_amdgcn_readlane(cell, __builtin_ctzll(todo));
                    todo &= todo - 1;
                }
                return true;
            };
            auto issue = [&](float (&tt)[8][NJ], int ca, int cb) {
                const float* __restrict__ spa = src + (size_t)ca * C;
                const float* __restrict__ spb = src + (size_t)cb * C;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    tt[0][j] = spa[j]; tt[1][j] = spa[C + j];
                    tt[2][j] = spa[(size_t)W2 * C + j]; tt[3][j] = spa[(size_t)W2 * C + C + j];
                    tt[4][j] = spb[j]; tt[5][j] = spb[C + j];
                    tt[6][j] = spb[(size_t)W2 * C + j]; tt[7][j] = spb[(size_t)W2 * C + C + j];
                }
            };
            auto consume = [&](const float (&tt)[8][NJ], int ca, int cb) {
                float d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int k = 0; k < 8; ++k) d[k] = fmaf(kf[j], tt[k][j], d[k]);
                wave_sum8(d, lane);
                // corr = sum_taps w_tap * <f_key, f_src(tap)>, taps in grid_sample's order nw, ne, sw, se
                const float blend_a = fmaf(d[3], t.w[3], fmaf(d[2], t.w[2], fmaf(d[1], t.w[1], d[0] * t.w[0])));
                const float blend_b = fmaf(d[7], t.w[3], fmaf(d[6], t.w[2], fmaf(d[5], t.w[1], d[4] * t.w[0])));
                acc = cell == ca ? blend_a : cell == cb ? blend_b : acc;
            };
            bool more = pick(ca0, cb0);
            if (more) issue(ta, ca0, cb0);
            while (more) {  // wave-uniform
                const bool nxt = pick(ca1, cb1);
                if (nxt) issue(tb, ca1, cb1);
                consume(ta, ca0, cb0);
                if (!nxt) break;
                more = pick(ca0, cb0);
                if (more) issue(ta, ca0, cb0);
                consume(tb, ca1, cb1);
            }
            if (live) {
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
    if (p.out_ps > 0) {  // pixel-major: the S values of a pixel are contiguous (what the 2-D convolutions after the sweep read)
        float* __restrict__ cn = p.corr.p[v] + (((size_t)n * h + y) * w + x0) * p.out_ps;
        float* __restrict__ mn = p.mask.p[v] + (((size_t)n * h + y) * w + x0) * p.out_ps;
        for (int e = tid; e < S * SWEEP_PX; e += 256) {
            const int px = e / S, s = e % S;
            if (px < npx) {
                cn[(size_t)px * p.out_ps + s] = res[s * SWEEP_PX + px];
                mn[(size_t)px * p.out_ps + s] = res[(S + s) * SWEEP_PX + px];
            }
        }
        return;
    }
    for (int e = tid; e < S * SWEEP_PX; e += 256) {
        const int s = e / SWEEP_PX, px = e % SWEEP_PX;
        if (px < npx) {
            co[(size_t)s * plane + px] = res[e];
            mo[(size_t)s * plane + px] = res[S * SWEEP_PX + e];
        }
    }
}

// ---- K1, second form: every source pixel's dot product with the key feature is computed ONCE per 64 planes -------------------------
// The kernel above takes the 2 x 2 cell of every distinct sampling position on its own: neighbouring cells along the epipolar line
// share two of their four pixels, so nearly half of its gathers and dot products are repeats (profiles/k1_pmc.json: texture
// addresser 65 % busy, 13.6 GB through the L1 for 184 MB of algorithmic input).  Here, per key pixel (one wave) and 64 planes:
//   1. lanes = planes: sampling position, bilinear weights, mask, cell (as above; same arithmetic, same values);
//   2. the distinct cells, in ray order, are compacted (ballot + prefix count) so that lanes = cells;
//   3. every cell decides in closed form which of its 4 pixels the one or two cells before it already hold (a pixel lies in at most
//      three consecutive cells of a monotone path; a longer chain just loads it again) and the new pixels get consecutive places
//      in a list (wave prefix sum);
//   4. the list is worked off 8 pixels per step: lane = (pixel l / 8, channel group l % 8), C / 8 channels per lane in 16-byte pieces
//      128 bytes apart (each load instruction reads whole 128-byte lines: with C / 8 contiguous channels per lane it touched 64 lines
//      and ran half as fast), 3 xor-steps finish a dot product; no per-cell control flow, loads one step ahead;
//   5. lanes = planes again: 4 dot products by index, blended with the weights in grid_sample's order (nw, ne, sw, se).
// Dot products are summed in a different order than above (rounding differs in the last bits); masks are identical.
#define MVD_WAVE_LDS_SYNC()                                     \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)

constexpr int SWEEP_SCR_INTS = 64 + 256 + 256 + 256;  // per wave: compacted cells | pixel list | 4 list places per cell | dot products

template <int NCH>  // C = 8 * NCH
__global__ void __launch_bounds__(256) sweep_corr_px_kernel(SweepParams p) {
    extern __shared__ __attribute__((aligned(16))) float res[];  // [2][S][SWEEP_PX] | 4 x per-wave scratch
    constexpr int C = 8 * NCH;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    int u = (int)(blockIdx.x % 8) * p.per_xcd + (int)(blockIdx.x / 8);
    if (p.per_xcd == 0) u = blockIdx.x;
    if (u >= p.total) return;
    const int xt = u % p.tiles_x; u /= p.tiles_x;
    const int y = u % p.h; u /= p.h;
    const int v = u % p.V;
    const int n = u / p.V;
    const int x0 = xt * SWEEP_PX;
    const int h = p.h, w = p.w, hs = p.hs, ws = p.ws, S = p.S;
    const int W2 = ws + 3;

    int* const scr = reinterpret_cast<int*>(res + 2 * S * SWEEP_PX) + wave * SWEEP_SCR_INTS;
    int* const cellbuf = scr;
    int* const pixlist = scr + 64;
    int* const cellpix = scr + 64 + 256;
    float* const dots = reinterpret_cast<float*>(scr + 64 + 512);

    const Epi E = epipolar(p.K_key + n * 9, p.K_src.p[v] + n * 9, p.T.p[v] + n * 16, h, w, hs, ws);
    const int cg = lane & 7, slot = lane >> 3;
    // channels of lane group cg: 4 (cg + 8 j) .. + 3, j = 0 .. NCH / 4 - 1: load j of a pixel's 8 lanes is one contiguous 128-byte line
    const float* __restrict__ src = p.src.p[v] + (size_t)n * (hs + 3) * W2 * C + cg * 4;
    const float* __restrict__ invd = p.invd + (size_t)n * p.invd_stride;
    const float inv_sqrt_c = p.corr_scale;
    const float fws = (float)ws, fhs = (float)hs;
    const float yc = (float)y + 0.5f;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    for (int pi = wave; pi < SWEEP_PX; pi += 4) {
        const int x = x0 + pi;
        if (x >= w) break;  // wave-uniform
        const float xc = (float)x + 0.5f;
        const float u_inf = (E.a * xc + E.b * yc) + E.c;
        const float v_inf = (E.f * xc + E.g * yc) + E.h;
        const float k_inf = (E.j * xc + E.k * yc) + E.l;
        const float z_pole = -(E.m / k_inf);

        float kf[NCH];
        const float* kp = p.key + (((size_t)n * h + y) * w + x) * C + cg * 4;
#pragma unroll
        for (int j = 0; j < NCH; j += 4) {
            const float4 t4 = *reinterpret_cast<const float4*>(kp + 8 * j);
            kf[j] = t4.x; kf[j + 1] = t4.y; kf[j + 2] = t4.z; kf[j + 3] = t4.w;
        }

        for (int s0 = 0; s0 < S; s0 += 64) {
            // ---- 1. geometry of plane s0 + lane (the arithmetic of sweep_corr_kernel) ----
            const int s = s0 + lane;
            const bool live = s < S;
            const int sc = live ? s : S - 1;
            const float ds = p.invd_per_pixel ? p.invd[(((size_t)n * S + sc) * h + y) * w + x] : invd[sc];
            const float den = k_inf + E.m * ds;
            const float us = replace_nonfinite((u_inf + E.e * ds) / den);
            const float vs = replace_nonfinite((v_inf + E.i * ds) / den);
            const float zs = 1.0f / ds;
            const bool visible = (zs > 0.f) && (((k_inf > 0.f) && (zs > z_pole)) || ((k_inf < 0.f) && (zs < z_pole)) ||
                                               ((k_inf == 0.f) && (E.m > 0.f)));
            const float ix = unnormalize_coord(2.0f * us / fws - 1.0f, fws);
            const float iy = unnormalize_coord(2.0f * vs / fhs - 1.0f, fhs);
            const Taps t = bilinear_taps(ix, iy, hs, ws);
            const float mk = (t.inb < 0.9999f || !visible) ? 0.f : 1.f;
            const int cx = (int)fminf(fmaxf(floorf(ix), -1.0f), (float)(ws - 1)) + 1;
            const int cy = (int)fminf(fmaxf(floorf(iy), -1.0f), (float)(hs - 1)) + 1;
            const int cell = live ? cy * W2 + cx : -1;

            // ---- 2. distinct cells in ray order -> lanes ----
            const int prev = __shfl_up(cell, 1);
            const bool first = live && (lane == 0 || cell != prev);
            const unsigned long long fb = __ballot(first);
            const int ncell = __popcll(fb);
            const int dci = __popcll(fb & lt_mask) + (first ? 1 : 0) - 1;  // index of this plane's cell among the distinct ones
            if (first) cellbuf[dci] = cell;
            MVD_WAVE_LDS_SYNC();
            const bool isc = lane < ncell;
            const int cj = isc ? cellbuf[lane] : 0x3fffffff;

            // ---- 3. which pixels are new, and where every pixel's dot product will be ----
            const int c1 = __shfl_up(cj, 1), c2 = __shfl_up(cj, 2), c3 = __shfl_up(cj, 3);
            const int offs[4] = {0, 1, W2, W2 + 1};
            auto in_cell = [W2](int d) { return d == 0 || d == 1 || d == W2 || d == W2 + 1; };
            auto pos_of = [W2](int d) { return d == 0 ? 0 : d == 1 ? 1 : d == W2 ? 2 : 3; };
            unsigned newmask = 0;
            int owner[4], opos[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = cj + offs[k];
                const bool in1 = lane >= 1 && in_cell(q - c1);
                const bool in2 = in1 && lane >= 2 && in_cell(q - c2);
                const bool in3 = in2 && lane >= 3 && in_cell(q - c3);
                const bool isnew = !in1 || in3;  // (a chain of four cells through one pixel: load it again)
                newmask |= isnew ? (1u << k) : 0u;
                owner[k] = isnew ? 0 : in2 ? 2 : 1;
                opos[k] = pos_of(in2 ? q - c2 : q - c1);
            }
            if (!isc) newmask = 0;
            const int newcnt = __popc(newmask);
            int incl = newcnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            const int base = incl - newcnt;
            const int npix = __builtin_amdgcn_readlane(incl, 63);
            const int b1 = __shfl_up(base, 1), b2 = __shfl_up(base, 2);
            const unsigned m1 = (unsigned)__shfl_up((int)newmask, 1), m2 = (unsigned)__shfl_up((int)newmask, 2);
            int idx[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ob = owner[k] == 0 ? base : owner[k] == 1 ? b1 : b2;
                const unsigned om = owner[k] == 0 ? newmask : owner[k] == 1 ? m1 : m2;
                const int op = owner[k] == 0 ? k : opos[k];
                idx[k] = ob + __popc(om & ((1u << op) - 1u));
                if (isc && owner[k] == 0) pixlist[idx[k]] = cj + offs[k];
            }
            if (isc) *reinterpret_cast<int4*>(cellpix + 4 * lane) = make_int4(idx[0], idx[1], idx[2], idx[3]);
            MVD_WAVE_LDS_SYNC();

            // ---- 4. the dot products: 8 pixels per step, 8 lanes per pixel ----
            const int nsteps = (npix + 7) >> 3;  // wave-uniform, >= 1
            float ta[NCH], tb[NCH];
            auto issue = [&](int i, float (&tt)[NCH]) {
                const int li = min(8 * i + slot, npix - 1);
                const float* __restrict__ sp = src + (size_t)pixlist[li] * C;
#pragma unroll
                for (int j = 0; j < NCH; j += 4) {
                    const float4 t4 = *reinterpret_cast<const float4*>(sp + 8 * j);
                    tt[j] = t4.x; tt[j + 1] = t4.y; tt[j + 2] = t4.z; tt[j + 3] = t4.w;
                }
            };
            auto consume = [&](int i, const float (&tt)[NCH]) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int j = 0; j < NCH; j += 4) {
                    a0 = fmaf(kf[j], tt[j], a0); a1 = fmaf(kf[j + 1], tt[j + 1], a1);
                    a2 = fmaf(kf[j + 2], tt[j + 2], a2); a3 = fmaf(kf[j + 3], tt[j + 3], a3);
                }
                float d = (a0 + a1) + (a2 + a3);
                d += __shfl_xor(d, 1);
                d += __shfl_xor(d, 2);
                d += __shfl_xor(d, 4);
                if (cg == 0 && 8 * i + slot < npix) dots[8 * i + slot] = d;
            };
            issue(0, ta);
            for (int i = 0; i < nsteps; i += 2) {  // wave-uniform
                if (i + 1 < nsteps) issue(i + 1, tb);
                consume(i, ta);
                if (i + 1 >= nsteps) break;
                if (i + 2 < nsteps) issue(i + 2, ta);
                consume(i + 1, tb);
            }
            MVD_WAVE_LDS_SYNC();

            // ---- 5. planes: blend the four dot products of the plane's cell ----
            if (live) {
                const int4 li = *reinterpret_cast<const int4*>(cellpix + 4 * dci);
                const float d0 = dots[li.x], d1 = dots[li.y], d2 = dots[li.z], d3 = dots[li.w];
                const float acc = fmaf(d3, t.w[3], fmaf(d2, t.w[2], fmaf(d1, t.w[1], d0 * t.w[0])));
                res[s * SWEEP_PX + pi] = acc * inv_sqrt_c * mk;
                res[(S + s) * SWEEP_PX + pi] = mk;
            }
            MVD_WAVE_LDS_SYNC();  // the scratch is rewritten by the next 64 planes
        }
    }
    __syncthreads();
    const int npx = min(SWEEP_PX, w - x0);
    const size_t plane = (size_t)h * w;
    if (p.out_ps > 0) {
        float* __restrict__ cn = p.corr.p[v] + (((size_t)n * h + y) * w + x0) * p.out_ps;
        float* __restrict__ mn = p.mask.p[v] + (((size_t)n * h + y) * w + x0) * p.out_ps;
        float am = 0.f;
        for (int e = tid; e < S * SWEEP_PX; e += 256) {
            const int px = e / S, s = e % S;
            if (px < npx) {
                const float c = res[s * SWEEP_PX + px];
                cn[(size_t)px * p.out_ps + s] = c;
                mn[(size_t)px * p.out_ps + s] = res[(S + s) * SWEEP_PX + px];
                am = fmaxf(am, finite_abs_or_zero(c));
            }
        }
        if (p.amax) {  // what the fusion block's first convolution scales its activations by; a wave rarely has to raise the slot
            for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
            if (lane == 0) raise_absmax(p.amax, am);
        }
        return;
    }
    float* __restrict__ co = p.corr.p[v] + ((size_t)n * S * h + y) * w + x0;
    float* __restrict__ mo = p.mask.p[v] + ((size_t)n * S * h + y) * w + x0;
    for (int e = tid; e < S * SWEEP_PX; e += 256) {
        const int s = e / SWEEP_PX, px = e % SWEEP_PX;
        if (px < npx) {
            co[(size_t)s * plane + px] = res[e];
            mo[(size_t)s * plane + px] = res[S * SWEEP_PX + e];
        }
    }
}

// WarpOnlyCorr (planesweep_corr.py:107-140): the plane sweep without the correlation — the source features sampled at the
// S positions of every key pixel, times the sampling mask.  Same grids and mask as sweep_corr_kernel; source features in the
// caller's own (N,C,hs,ws) layout (one thread per (plane, key pixel) walks the channels, so neighbouring threads read
// neighbouring source pixels of one channel plane and write neighbouring outputs).  norm_after: normalize(warped, dim=C)
// = x / (|x|_2 + 1e-9) (:8-10,135-136) before the mask is applied.  Output (N,S,C,h,w), mask (N,S,h,w).
struct WarpOnlyParams {
    ViewPtrs src;    // V x (N,C,hs,ws)
    ViewPtrs K_src;  // V x (N,3,3)
    ViewPtrs T;      // V x (N,4,4)
    ViewOutPtrs out; // V x (N,S,C,h,w)
    ViewOutPtrs mask;
    const float* K_key;
    const float* invd;
    int invd_stride, invd_per_pixel, norm_after;
    int N, C, h, w, hs, ws, S, V;
};

__global__ void __launch_bounds__(256) sweep_warp_kernel(WarpOnlyParams p) {
    const int h = p.h, w = p.w, hs = p.hs, ws = p.ws, S = p.S, C = p.C;
    const int v = blockIdx.z % p.V, n = blockIdx.z / p.V;
    const int s = blockIdx.y;
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= (long long)h * w) return;
    const int y = (int)(pix / w), x = (int)(pix - (long long)y * w);
    const Epi E = epipolar(p.K_key + n * 9, p.K_src.p[v] + n * 9, p.T.p[v] + n * 16, h, w, hs, ws);
    const float xc = (float)x + 0.5f, yc = (float)y + 0.5f;
    const float u_inf = (E.a * xc + E.b * yc) + E.c;
    const float v_inf = (E.f * xc + E.g * yc) + E.h;
    const float k_inf = (E.j * xc + E.k * yc) + E.l;
    const float z_pole = -(E.m / k_inf);
    const float ds = p.invd_per_pixel ? p.invd[(((size_t)n * S + s) * h + y) * w + x] : p.invd[(size_t)n * p.invd_stride + s];
    const float den = k_inf + E.m * ds;
    const float us = replace_nonfinite((u_inf + E.e * ds) / den);
    const float vs = replace_nonfinite((v_inf + E.i * ds) / den);
    const float zs = 1.0f / ds;
    const bool visible = (zs > 0.f) && (((k_inf > 0.f) && (zs > z_pole)) || ((k_inf < 0.f) && (zs < z_pole)) ||
                                       ((k_inf == 0.f) && (E.m > 0.f)));
    const float fws = (float)ws, fhs = (float)hs;
    const float ix = unnormalize_coord(2.0f * us / fws - 1.0f, fws);
    const float iy = unnormalize_coord(2.0f * vs / fhs - 1.0f, fhs);
    const Taps t = bilinear_taps(ix, iy, hs, ws);
    // WarpOnlyCorr does not take the visibility mask (its forward is called with grids only, :131-133, the `mask`
    // argument is unused): the returned mask is the sampling mask alone
    (void)visible;
    const float mk = t.inb < 0.9999f ? 0.f : 1.f;
    const size_t plane = (size_t)hs * ws;
    const float* __restrict__ sp = p.src.p[v] + (size_t)n * C * plane;
    float scale = mk;
    if (p.norm_after) {
        float ss = 0.f;
        for (int c = 0; c < C; ++c) {
            const float* q = sp + (size_t)c * plane;
            const float val = fmaf(q[t.off[3]], t.w[3], fmaf(q[t.off[2]], t.w[2], fmaf(q[t.off[1]], t.w[1], q[t.off[0]] * t.w[0])));
            ss = fmaf(val, val, ss);
        }
        scale = mk / (sqrtf(ss) + 1e-9f);
    }
    float* __restrict__ op = p.out.p[v] + (((size_t)n * S + s) * C) * ((size_t)h * w) + pix;
    for (int c = 0; c < C; ++c) {
        const float* q = sp + (size_t)c * plane;
        const float val = fmaf(q[t.off[3]], t.w[3], fmaf(q[t.off[2]], t.w[2], fmaf(q[t.off[1]], t.w[1], q[t.off[0]] * t.w[0])));
        op[(size_t)c * h * w] = val * scale;
    }
    p.mask.p[v][((size_t)n * S + s) * ((size_t)h * w) + pix] = mk;
}

int transpose_launch(const float* src, float* dst, int N, long long rows, long long cols, hipStream_t st);
int repack_padded_launch(const float* src, float* dst, int B, int C, int h, int w, hipStream_t st);
size_t padded_slot_bytes_public(int B, int C, int h, int w);

}  // namespace mvd

extern "C" {

size_t mvd_sweep_corr_workspace_bytes(int N, int C, int h, int w, int hs, int ws, int V) {
    if (N <= 0 || C <= 0 || h <= 0 || w <= 0 || hs <= 0 || ws <= 0 || V <= 0) return 0;
    // channel-last key copy + V zero-bordered channel-last source copies ((hs+3) x (ws+3) pixels each)
    return mvd::align_up((size_t)N * C * h * w * sizeof(float), 256) + (size_t)V * mvd::padded_slot_bytes_public(N, C, hs, ws);
}

int mvd_sweep_corr_f32(const float* feat_key, const float* const* feat_src, const float* K_key,
                       const float* const* K_src, const float* const* T_src2key, const float* invdepths,
                       int invdepth_batched, int N, int C, int h, int w, int hs, int ws, int S, int V,
                       float* const* corr_out, float* const* mask_out, void* workspace, size_t workspace_bytes,
                       mvd_stream_t stream) {
    return mvd_sweep_corr_ex_f32(feat_key, feat_src, K_key, K_src, T_src2key, invdepths,
                                 invdepth_batched ? MVD_INVDEPTH_BATCHED : MVD_INVDEPTH_SHARED, 1.0f / sqrtf((float)(C > 0 ? C : 1)), N,
                                 C, h, w, hs, ws, S, V, corr_out, mask_out, workspace, workspace_bytes, stream);
}

static int sweep_corr_run(const float* key_nhwc, const float* const* src_bordered, const float* K_key, const float* const* K_src,
                          const float* const* T_src2key, const float* invdepths, int invdepth_mode, float corr_scale, int N, int C, int h,
                          int w, int hs, int ws, int S, int V, float* const* corr_out, float* const* mask_out, int out_ps, float* amax,
                          hipStream_t st) {
    const size_t lds = (size_t)2 * S * mvd::SWEEP_PX * sizeof(float);
    mvd::SweepParams p{};
    p.key = key_nhwc;
    for (int v = 0; v < V; ++v) {
        p.src.p[v] = src_bordered[v];
        p.K_src.p[v] = K_src[v];
        p.T.p[v] = T_src2key[v];
        p.corr.p[v] = corr_out[v];
        p.mask.p[v] = mask_out[v];
    }
    p.K_key = K_key;
    p.invd = invdepths;
    p.invd_stride = invdepth_mode == MVD_INVDEPTH_BATCHED ? S : 0;
    p.invd_per_pixel = invdepth_mode == MVD_INVDEPTH_PER_PIXEL;
    p.corr_scale = corr_scale;
    p.N = N; p.h = h; p.w = w; p.hs = hs; p.ws = ws; p.S = S; p.V = V;
    p.out_ps = out_ps;
    p.amax = amax;
    p.tiles_x = (w + mvd::SWEEP_PX - 1) / mvd::SWEEP_PX;
    const long long total = (long long)p.tiles_x * h * N * V;
    if (total > 0x7fffff00LL) {
        mvd::set_error("sweep_corr: %lld workgroups exceed the grid limit", total);
        return MVD_ERR_INVALID_ARG;
    }
    p.total = (int)total;
    p.per_xcd = mvd::exp_env("MVD_K1_XCD_ORDER") ? (p.total + 7) / 8 : 0;
    dim3 grid((unsigned)(p.per_xcd ? 8 * p.per_xcd : p.total));
    mvd::timing_begin(st);
    if (C <= 256 && !mvd::exp_env("MVD_K1_CELLS")) {  // every source pixel once per 64 planes (sweep_corr_px_kernel)
        const size_t lds2 = lds + (size_t)4 * mvd::SWEEP_SCR_INTS * sizeof(int);
        if (lds2 > 160 * 1024) {
            mvd::set_error("sweep_corr: S=%d needs %zu B of LDS (> 160 KiB)", S, lds2);
            return MVD_ERR_INVALID_ARG;
        }
        switch (C / 64) {
#define MVD_CASE(NJ)                                                                                                  \
    case NJ:                                                                                                          \
        if (lds2 > 64 * 1024 &&                                                                                       \
            hipFuncSetAttribute((const void*)mvd::sweep_corr_px_kernel<8 * NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds2) != hipSuccess)                                                             \
            return mvd::launch_status("sweep_corr: LDS attribute");                                                   \
        hipLaunchKernelGGL((mvd::sweep_corr_px_kernel<8 * NJ>), grid, dim3(256), lds2, st, p);                       \
        break;
            MVD_CASE(1) MVD_CASE(2) MVD_CASE(3) MVD_CASE(4)
#undef MVD_CASE
        }
        mvd::timing_end(st);
        return mvd::launch_status("sweep_corr");
    }
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

static int sweep_corr_check(const void* feat_key, const void* feat_src, const void* K_key, const void* K_src, const void* T_src2key,
                            const void* invdepths, int invdepth_mode, int N, int C, int h, int w, int hs, int ws, int S, int V,
                            const void* corr_out, const void* mask_out) {
    MVD_REQUIRE(invdepth_mode == MVD_INVDEPTH_SHARED || invdepth_mode == MVD_INVDEPTH_BATCHED || invdepth_mode == MVD_INVDEPTH_PER_PIXEL,
                "sweep_corr: invdepth_mode %d", invdepth_mode);
    MVD_REQUIRE(feat_key && feat_src && K_key && K_src && T_src2key && invdepths && corr_out && mask_out,
                "sweep_corr: NULL argument");
    MVD_REQUIRE(N > 0 && h > 0 && w > 0 && hs > 0 && ws > 0 && S > 0, "sweep_corr: non-positive dimension");
    MVD_REQUIRE(V >= 1 && V <= MVD_MAX_VIEWS, "sweep_corr: V=%d outside 1..%d", V, MVD_MAX_VIEWS);
    MVD_REQUIRE(C % 64 == 0 && C >= 64 && C <= 512, "sweep_corr: C=%d must be a multiple of 64 in 64..512", C);
    MVD_REQUIRE((long long)(hs + 3) * (ws + 3) * C < 0x7fffffffLL, "sweep_corr: source map %dx%dx%d too large", hs, ws, C);
    MVD_REQUIRE(h <= 65535 && (long long)N * V <= 65535, "sweep_corr: h or N*V exceeds 65535");
    const size_t lds = (size_t)2 * S * mvd::SWEEP_PX * sizeof(float);
    MVD_REQUIRE(lds <= 160 * 1024, "sweep_corr: S=%d needs %zu B of LDS (> 160 KiB)", S, lds);
    return MVD_OK;
}

int mvd_sweep_corr_ex_f32(const float* feat_key, const float* const* feat_src, const float* K_key,
                          const float* const* K_src, const float* const* T_src2key, const float* invdepths,
                          int invdepth_mode, float corr_scale, int N, int C, int h, int w, int hs, int ws, int S, int V,
                          float* const* corr_out, float* const* mask_out, void* workspace, size_t workspace_bytes,
                          mvd_stream_t stream) {
    int rc = sweep_corr_check(feat_key, feat_src, K_key, K_src, T_src2key, invdepths, invdepth_mode, N, C, h, w, hs, ws, S, V, corr_out, mask_out);
    if (rc) return rc;
    const size_t need = mvd_sweep_corr_workspace_bytes(N, C, h, w, hs, ws, V);
    if (!workspace || workspace_bytes < need) {
        mvd::set_error("sweep_corr: workspace %zu B < required %zu B", workspace_bytes, need);
        return MVD_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* wsp = (float*)workspace;
    rc = mvd::transpose_launch(feat_key, wsp, N, C, (long long)h * w, st);
    if (rc) return rc;
    const float* key = wsp;
    wsp += mvd::align_up((size_t)N * C * h * w * sizeof(float), 256) / sizeof(float);
    const size_t per = mvd::padded_slot_bytes_public(N, C, hs, ws) / sizeof(float);
    const float* srcs[MVD_MAX_VIEWS];
    for (int v = 0; v < V; ++v) {
        MVD_REQUIRE(feat_src[v] && K_src[v] && T_src2key[v] && corr_out[v] && mask_out[v], "sweep_corr: NULL view %d", v);
        rc = mvd::repack_padded_launch(feat_src[v], wsp + v * per, N, C, hs, ws, st);
        if (rc) return rc;
        srcs[v] = wsp + v * per;
    }
    return sweep_corr_run(key, srcs, K_key, K_src, T_src2key, invdepths, invdepth_mode, corr_scale, N, C, h, w, hs, ws, S, V, corr_out, mask_out, 0, nullptr, st);
}

int mvd_sweep_corr_nhwc_f32(const float* feat_key, const float* const* feat_src, const float* K_key, const float* const* K_src,
                            const float* const* T_src2key, const float* invdepths, int invdepth_mode, float corr_scale, int N, int C,
                            int h, int w, int hs, int ws, int S, int V, float* const* corr_out, float* const* mask_out,
                            int out_pixel_stride, float* corr_absmax, mvd_stream_t stream) {
    int rc = sweep_corr_check(feat_key, feat_src, K_key, K_src, T_src2key, invdepths, invdepth_mode, N, C, h, w, hs, ws, S, V, corr_out, mask_out);
    if (rc) return rc;
    MVD_REQUIRE(out_pixel_stride >= S, "sweep_corr_nhwc: output pixel stride %d below S=%d", out_pixel_stride, S);
    for (int v = 0; v < V; ++v)
        MVD_REQUIRE(feat_src[v] && K_src[v] && T_src2key[v] && corr_out[v] && mask_out[v], "sweep_corr: NULL view %d", v);
    MVD_REQUIRE(!corr_absmax || C <= 256, "sweep_corr_nhwc: corr_absmax is built for C <= 256");
    return sweep_corr_run(feat_key, feat_src, K_key, K_src, T_src2key, invdepths, invdepth_mode, corr_scale, N, C, h, w, hs, ws, S, V, corr_out,
                          mask_out, out_pixel_stride, corr_absmax, (hipStream_t)stream);
}
int mvd_sweep_warp_f32(const float* const* feat_src, const float* K_key, const float* const* K_src,
                       const float* const* T_src2key, const float* invdepths, int invdepth_mode, int normalize_after, int N,
                       int C, int h, int w, int hs, int ws, int S, int V, float* const* warped_out, float* const* mask_out,
                       mvd_stream_t stream) {
    MVD_REQUIRE(invdepth_mode == MVD_INVDEPTH_SHARED || invdepth_mode == MVD_INVDEPTH_BATCHED || invdepth_mode == MVD_INVDEPTH_PER_PIXEL,
                "sweep_warp: invdepth_mode %d", invdepth_mode);
    MVD_REQUIRE(feat_src && K_key && K_src && T_src2key && invdepths && warped_out && mask_out, "sweep_warp: NULL argument");
    MVD_REQUIRE(N > 0 && C > 0 && h > 0 && w > 0 && hs > 0 && ws > 0 && S > 0, "sweep_warp: non-positive dimension");
    MVD_REQUIRE(V >= 1 && V <= MVD_MAX_VIEWS, "sweep_warp: V=%d outside 1..%d", V, MVD_MAX_VIEWS);
    MVD_REQUIRE(S <= 65535 && (long long)N * V <= 65535, "sweep_warp: S or N*V exceeds 65535");
    MVD_REQUIRE((long long)hs * ws < 0x7fffffffLL, "sweep_warp: source map %dx%d too large", hs, ws);
    mvd::WarpOnlyParams p{};
    for (int v = 0; v < V; ++v) {
        MVD_REQUIRE(feat_src[v] && K_src[v] && T_src2key[v] && warped_out[v] && mask_out[v], "sweep_warp: NULL view %d", v);
        p.src.p[v] = feat_src[v]; p.K_src.p[v] = K_src[v]; p.T.p[v] = T_src2key[v];
        p.out.p[v] = warped_out[v]; p.mask.p[v] = mask_out[v];
    }
    p.K_key = K_key; p.invd = invdepths;
    p.invd_stride = invdepth_mode == MVD_INVDEPTH_BATCHED ? S : 0;
    p.invd_per_pixel = invdepth_mode == MVD_INVDEPTH_PER_PIXEL;
    p.norm_after = normalize_after ? 1 : 0;
    p.N = N; p.C = C; p.h = h; p.w = w; p.hs = hs; p.ws = ws; p.S = S; p.V = V;
    const long long npix = (long long)h * w;
    dim3 grid((unsigned)((npix + 255) / 256), (unsigned)S, (unsigned)(N * V));
    hipLaunchKernelGGL(mvd::sweep_warp_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return mvd::launch_status("sweep_warp");
}
}
