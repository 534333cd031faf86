// K4 — one CostRegNet layer (rmvd/models/blocks/mvsnet_components.py:69-123) as an implicit GEMM on
// the gfx950 fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 fmaf chains, so the result is
// fp32-faithful to the reference's CPU path): 3x3x3 Conv3d stride 1/2 or ConvTranspose3d stride 2,
// with the eval-mode BatchNorm folded into a per-channel scale/shift, optional ReLU and optional skip
// addition fused into the epilogue.  Activations are channel-last (B,D,h,w,C).
//
// GEMM view per kernel tap: D[cout, voxel] += W_tap[cout, cin] * X[cin, voxel + tap].
//   A operand = weights (M = 16 couts), pre-packed in fragment order and read straight from L1/L2;
//   B operand = activations (N = 16 consecutive output voxels of one row), read from an LDS slab that
//               holds the input rows/halo of the current input depth plane;
//   K         = input channels, consumed in groups of KG = min(Cin, 16): one ds_read of R = KG/4
//               consecutive channels per lane feeds R MFMAs (MFMA j of a group covers channels
//               {R*q + j : q = 0..3}; the weight packing uses the same permutation).
// The C/D layout puts 4 consecutive couts of one voxel in each lane, so the epilogue stores float4s
// that tile the channel-last output contiguously.
//
// Workgroup = 4 waves; wave r owns output row r of a TH=4 x TW=16*MT tile of one output depth plane
// (for the transposed conv: of one output (d,h)-parity class and BOTH w-parities, which turns it into
// small dense convs whose results interleave into full contiguous output rows).
//
// Cout = 8 (conv0, the layer that carries 2/3 of the regulariser's FLOPs) would leave half of the 16 MFMA
// rows empty.  Its stride-1 form is therefore run in PAIR mode: the 16 rows are (w-phase j in {0,1}) x
// (8 couts) and the columns are PAIRS of adjacent output voxels, i.e. a stride-2-in-w convolution with a
// 4-tap kernel W'[j][t] = W[t-j] (zero outside 0..2): 4/3 of the taps but twice the useful rows, 1.5x
// fewer MFMAs for the same result (each product that is kept is bit-identical; the extra ones multiply by 0).
// Cout = 1 (the final `prob` layer) would use 1 row of 16: it has its own vector-ALU kernel below.
#include "mvd_common.h"
#include <stdlib.h>

namespace mvd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));

constexpr int CONV_TH = 4;
constexpr int CONV_PAD = 4;  // floats of padding per LDS pixel (keeps 16-B alignment, spreads banks)

struct ConvParams {
    const float* x;
    const float* wpk;
    const float* scale;
    const float* shift;
    const float* skip;
    float* y;
    float* absmax;       // optional (device, one float, zeroed by the caller): max |y| over the finite outputs (conv3d_kernel only)
    int B, Di, hi, wi;   // input dims
    int Do, ho, wo;      // output dims
    int Cout;
    int relu;
    int tiles_w, tiles_h;  // tiles over the (per-parity) output grid
    int per_xcd;           // experiments (MVD_K4_XCD): > 0 = XCD k takes the k-th run of per_xcd consecutive tiles
};

// launch index -> tile index (workgroups go to the 8 XCDs round-robin)
__device__ __forceinline__ int tile_index(const ConvParams& p) {
    const int bx = blockIdx.x;
    return (p.per_xcd > 0 && bx < 8 * p.per_xcd) ? (bx % 8) * p.per_xcd + bx / 8 : bx;
}
static int xcd_run(long long nblk, int bit) {
    const char* e = exp_env("MVD_K4_XCD");
    return (e && ((atoi(e) >> bit) & 1)) ? (int)(nblk / 8) : 0;
}

template <int CIN>
struct KGroup {
    static constexpr int KG = CIN >= 16 ? 16 : CIN;  // channels per k-group
    static constexpr int R = KG / 4;                 // consecutive channels per lane = MFMAs per group
    static constexpr int NKG = CIN / KG;
};

// packed weight index: [tap 27][k-group][n-tile][lane 64][R]
template <int CIN>
__host__ __device__ constexpr size_t packed_floats(int nt, int taps = 27) {
    return (size_t)taps * KGroup<CIN>::NKG * nt * 64 * KGroup<CIN>::R;
}
constexpr int MVD_CONV3D_S1_PAIR = 3;  // internal mode: stride-1 conv with Cout == 8 (see header)

template <int CIN>
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cout, int NT,
                                    int transposed) {
    using G = KGroup<CIN>;
    const bool pair = transposed == 2;  // PAIR mode: taps = (kd, kh, t in 0..3), row = phase * 8 + cout
    const bool dpair = transposed == 3;  // transposed conv with 8 output channels: taps = (kd, kh, s in 0..1), row = w-parity * 8 + cout
    const size_t total = packed_floats<CIN>(NT, pair ? 36 : dpair ? 18 : 27);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t r = e;
        const int j = r % G::R; r /= G::R;
        const int lane = r % 64; r /= 64;
        const int nt = r % NT; r /= NT;
        const int g = r % G::NKG; r /= G::NKG;
        const int tap = (int)r;
        const int row = nt * 16 + (lane & 15);
        const int cin = g * G::KG + G::R * (lane >> 4) + j;
        float val = 0.f;
        if (pair) {
            const int phase = row >> 3, cout = row & 7;
            const int t = tap & 3, kdh = tap >> 2;  // kdh = kd * 3 + kh
            const int kw = t - phase;
            if (kw >= 0 && kw <= 2) val = w[((size_t)cout * CIN + cin) * 27 + kdh * 3 + kw];
        } else if (dpair) {
            // step s reads input column c + s: s = 0 feeds out[2c] through kw = 1 and out[2c+1] through kw = 2,
            // s = 1 feeds out[2c+1] through kw = 0 (see deconv3d_pair_kernel)
            const int pw = row >> 3, cout = row & 7;
            const int st = tap & 1, kdh = tap >> 1;
            const int kw = st == 0 ? (pw ? 2 : 1) : (pw ? 0 : -1);
            if (kw >= 0) val = w[((size_t)cin * 8 + cout) * 27 + kdh * 3 + kw];  // ConvTranspose3d: (Cin,8,3,3,3)
        } else if (row < Cout) {
            val = transposed ? w[((size_t)cin * Cout + row) * 27 + tap]   // ConvTranspose3d: (Cin,Cout,3,3,3)
                             : w[((size_t)row * CIN + cin) * 27 + tap];   // Conv3d: (Cout,Cin,3,3,3)
        }
        packed[e] = val;
    }
}

// geometry of one workgroup tile per mode
template <int MODE, int MT>
struct TileGeom {
    static constexpr bool S2 = MODE == MVD_CONV3D_STRIDE2, DECONV = MODE == MVD_DECONV3D_STRIDE2, PAIR = MODE == MVD_CONV3D_S1_PAIR;
    static constexpr int TW = 16 * MT;  // GEMM columns per wave row: output voxels (S1/S2), voxel pairs (PAIR), input cols (DECONV)
    static constexpr int SX = (S2 || PAIR) ? 2 : 1;  // slab columns per GEMM column
    static constexpr int ROWS = S2 ? 2 * CONV_TH + 1 : DECONV ? CONV_TH + 1 : CONV_TH + 2;
    static constexpr int COLS = S2 ? 2 * TW + 1 : DECONV ? TW + 1 : PAIR ? 2 * TW + 2 : TW + 2;
    static constexpr int NPW = DECONV ? 2 : 1;       // output w-parity classes accumulated together
    static constexpr int NWT = DECONV ? 3 : PAIR ? 4 : 3;  // w-tap list length
};

// MODE: MVD_CONV3D_STRIDE1 / MVD_CONV3D_STRIDE2 / MVD_DECONV3D_STRIDE2 / MVD_CONV3D_S1_PAIR
template <int CIN, int NT, int MT, int MODE>
__global__ void __launch_bounds__(256) conv3d_kernel(ConvParams p) {
    const float amax_seen = absmax_seen(p.absmax);  // read now, used by the epilogue
    using G = KGroup<CIN>;
    using T = TileGeom<MODE, MT>;
    constexpr bool S2 = T::S2, DECONV = T::DECONV, PAIR = T::PAIR;
    constexpr int TW = T::TW, SX = T::SX, ROWS = T::ROWS, COLS = T::COLS, NPW = T::NPW;
    constexpr int PSTR = CIN + CONV_PAD;  // floats per LDS pixel
    extern __shared__ __attribute__((aligned(16))) float slab[];  // [ROWS][COLS][PSTR]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int vox = lane & 15, q = lane >> 4;

    // ---- which tile ---------------------------------------------------------------------------
    int bx = blockIdx.x;
    const int tw = bx % p.tiles_w; bx /= p.tiles_w;
    const int th = bx % p.tiles_h; bx /= p.tiles_h;
    int pd = 0, ph = 0;  // output (d, h) parity class (deconv only)
    if constexpr (DECONV) { pd = (bx >> 1) & 1; ph = bx & 1; bx >>= 2; }
    // grid z-extent: output depth planes (conv) or input depth planes (deconv, one per parity class)
    const int nzd = DECONV ? p.Di : p.Do;
    const int zd = bx % nzd;
    const int b = bx / nzd;
    const int r0 = th * CONV_TH;  // first tile row (output rows for conv, input rows for deconv)
    const int c0 = tw * TW;       // first GEMM column of the tile

    // input-plane list: conv: kd = 0..2 -> plane SZ*zd + kd - 1; deconv: pd=0: (k=1, plane zd); pd=1: (k=0, zd+1), (k=2, zd)
    constexpr int SZ = S2 ? 2 : 1;
    const int nplanes = DECONV ? (pd ? 2 : 1) : 3;
    // first input row / col held by the slab
    const int in_r0 = S2 ? 2 * r0 - 1 : DECONV ? r0 : r0 - 1;
    const int in_c0 = S2 ? 2 * c0 - 1 : DECONV ? c0 : PAIR ? 2 * c0 - 1 : c0 - 1;

    f32x4 acc[NPW][MT][NT];
#pragma unroll
    for (int c = 0; c < NPW; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[c][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this thread's share of a slab (float4 units; see conv3d_march_kernel): the next input plane is fetched into
    // registers before the current plane's MFMAs are issued and written to LDS after them
    constexpr int C4 = CIN / 4, NEL = ROWS * COLS * C4, NPF = (NEL + 255) / 256, DUMMY = ROWS * COLS * PSTR / 4;
    int loff[NPF], goff[NPF];
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
        const int e = tid + 256 * i;
        const int row = e / (COLS * C4), rem = e - row * (COLS * C4);
        const int col = rem / C4, c4 = rem - col * C4;
        const int gr = in_r0 + row, gc = in_c0 + col;
        loff[i] = e < NEL ? ((row * COLS + col) * PSTR) / 4 + c4 : DUMMY;
        goff[i] = (e < NEL && gr >= 0 && gr < p.hi && gc >= 0 && gc < p.wi) ? (gr * p.wi + gc) * C4 + c4 : -1;
    }
    const size_t plane_f4 = (size_t)p.hi * p.wi * C4;
    const float4* __restrict__ xb4 = reinterpret_cast<const float4*>(p.x) + (size_t)b * p.Di * plane_f4;
    float4* __restrict__ slab4 = reinterpret_cast<float4*>(slab);
    float4 pf[NPF];
    bool pf_ok = false;
    auto plane_of = [&](int ip) {
        if constexpr (DECONV) return pd ? (ip == 0 ? zd + 1 : zd) : zd;
        else return SZ * zd + ip - 1;
    };
    auto load_plane = [&](int plane) {
        pf_ok = plane >= 0 && plane < p.Di;  // block-uniform
        const float4* __restrict__ xp = xb4 + (size_t)(pf_ok ? plane : 0) * plane_f4;
#pragma unroll
        for (int i = 0; i < NPF; ++i) pf[i] = xp[max(goff[i], 0)];
    };
    load_plane(plane_of(0));

    for (int ip = 0; ip < nplanes; ++ip) {
        int kd;
        if constexpr (DECONV) kd = pd ? (ip == 0 ? 0 : 2) : 1;
        else kd = ip;
        const int plane = plane_of(ip);
        const bool plane_ok = plane >= 0 && plane < p.Di;  // block-uniform

        __syncthreads();  // previous plane's reads are done
        if (plane_ok) {
#pragma unroll
            for (int i = 0; i < NPF; ++i)
                slab4[loff[i]] = (pf_ok && goff[i] >= 0) ? pf[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        if (ip + 1 < nplanes) load_plane(plane_of(ip + 1));  // lands during this plane's MFMAs
        if (!plane_ok) continue;                              // an out-of-range plane contributes zeros

        // ---- taps of this plane ----------------------------------------------------------------
        const int nkh = DECONV ? (ph ? 2 : 1) : 3;
        for (int ih = 0; ih < nkh; ++ih) {
            int kh, srow;  // kernel index, slab row read by this wave
            if constexpr (DECONV) {
                kh = ph ? (ih == 0 ? 0 : 2) : 1;
                srow = wave + (ph ? (ih == 0 ? 1 : 0) : 0);
            } else {
                kh = ih;
                srow = (S2 ? 2 : 1) * wave + ih;
            }
            // weight fragments of this whole (kd, kh) step first, in one batch: vmcnt retires in order, so a load +
            // wait per tap would pay one L2 round trip per 4-16 MFMAs (when the batch is too big for the register
            // file it is split per w-tap)
            constexpr int dkw[3] = {1, 0, 2}, dsc[3] = {0, 1, 0}, dcl[3] = {0, 1, 1};
            constexpr bool ABATCH = T::NWT * G::NKG * NT * G::R <= 96;
            float afs[ABATCH ? T::NWT : 1][G::NKG][NT][G::R];
            auto load_a = [&](int iw, float (&dst)[G::NKG][NT][G::R]) {
                const int kw = DECONV ? dkw[iw % 3] : iw;
                const int tap = PAIR ? (kd * 3 + kh) * 4 + kw : (kd * 3 + kh) * 3 + kw;
#pragma unroll
                for (int g = 0; g < G::NKG; ++g)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const float* wp = p.wpk + ((((size_t)tap * G::NKG + g) * NT + n) * 64 + lane) * G::R;
                        if constexpr (G::R == 4) {
                            const float4 t = *reinterpret_cast<const float4*>(wp);
                            dst[g][n][0] = t.x; dst[g][n][1] = t.y; dst[g][n][2] = t.z; dst[g][n][3] = t.w;
                        } else {
                            const float2 t = *reinterpret_cast<const float2*>(wp);
                            dst[g][n][0] = t.x; dst[g][n][1] = t.y;
                        }
                    }
            };
            if constexpr (ABATCH) {
#pragma unroll
                for (int iw = 0; iw < T::NWT; ++iw) load_a(iw, afs[iw]);
            }
#pragma unroll
            for (int iw = 0; iw < T::NWT; ++iw) {
                // (slab column of GEMM column 0, output w-parity class)
                const int scol = DECONV ? dsc[iw % 3] : iw;
                const int cls = DECONV ? dcl[iw % 3] : 0;
                if constexpr (!ABATCH) load_a(iw, afs[0]);
                float (&af)[G::NKG][NT][G::R] = afs[ABATCH ? iw : 0];
                const float* __restrict__ srow_p = slab + (srow * COLS + scol) * PSTR + G::R * q;
#pragma unroll
                for (int g = 0; g < G::NKG; ++g) {
                    float bf[MT][G::R];
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const float* bp = srow_p + ((m * 16 + vox) * SX) * PSTR + g * G::KG;
                        if constexpr (G::R == 4) {
                            const float4 t = *reinterpret_cast<const float4*>(bp);
                            bf[m][0] = t.x; bf[m][1] = t.y; bf[m][2] = t.z; bf[m][3] = t.w;
                        } else {
                            const float2 t = *reinterpret_cast<const float2*>(bp);
                            bf[m][0] = t.x; bf[m][1] = t.y;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < G::R; ++j)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int n = 0; n < NT; ++n) {
                                if constexpr (NPW == 2) {
                                    if (cls == 0)
                                        acc[0][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g][n][j], bf[m][j], acc[0][m][n], 0, 0, 0);
                                    else
                                        acc[1][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g][n][j], bf[m][j], acc[1][m][n], 0, 0, 0);
                                } else {
                                    acc[0][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g][n][j], bf[m][j], acc[0][m][n], 0, 0, 0);
                                }
                            }
                }
            }
        }
    }

    // ---- epilogue: y = act(acc*scale + shift) (+ skip); lane holds rows 16n + 4q .. +3 of GEMM column `vox` ----
    float esc[NT][4], esh[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ch = (PAIR ? 4 * (q & 1) : n * 16 + q * 4) + k;
            esc[n][k] = ch < p.Cout ? p.scale[ch] : 0.f;
            esh[n][k] = ch < p.Cout ? p.shift[ch] : 0.f;
        }
    const int orow = DECONV ? 2 * (r0 + wave) + ph : r0 + wave;
    const int oz = DECONV ? 2 * zd + pd : zd;
    float amax = 0.f;  // max |y| over this lane's finite outputs (p.absmax)
    if (orow < p.ho) {
        const size_t row_base = (((size_t)b * p.Do + oz) * p.ho + orow) * p.wo;
#pragma unroll
        for (int c = 0; c < NPW; ++c)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int gcol = c0 + m * 16 + vox;
                const int ocol = DECONV ? 2 * gcol + c : PAIR ? 2 * gcol + (q >> 1) : gcol;
                if (ocol >= p.wo) continue;
                const size_t o = (row_base + ocol) * p.Cout;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int cb = PAIR ? 4 * (q & 1) : n * 16 + q * 4;  // first of this lane's 4 output channels
                    if (cb >= p.Cout) continue;
                    float r[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int ch = cb + k;
                        float val = 0.f;
                        if (ch < p.Cout) {
                            val = fmaf(acc[c][m][n][k], esc[n][k], esh[n][k]);
                            if (p.relu) val = fmaxf(val, 0.f);
                            if (p.skip) val += p.skip[o + ch];
                        }
                        r[k] = val;
                    }
                    if (p.absmax)
                        amax = fmaxf(fmaxf(amax, fmaxf(finite_abs_or_zero(r[0]), finite_abs_or_zero(r[1]))),
                                     fmaxf(finite_abs_or_zero(r[2]), finite_abs_or_zero(r[3])));
                    if (cb + 3 < p.Cout) {
                        *reinterpret_cast<float4*>(p.y + o + cb) = make_float4(r[0], r[1], r[2], r[3]);
                    } else {
                        for (int k = 0; k < 4; ++k)
                            if (cb + k < p.Cout) p.y[o + cb + k] = r[k];
                    }
                }
            }
    }
    if (p.absmax) {  // what the split-operand layer that consumes y scales by: ONE atomic per workgroup (they serialise on the address)
        __shared__ float wmax[4];
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        if (lane == 0) wmax[wave] = amax;
        __syncthreads();
        if (tid == 0) {
            amax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            raise_absmax_seen(p.absmax, amax, amax_seen);
        }
    }
}

// Stride-1 layers (plain and PAIR) in depth-marching form: a workgroup keeps its 4 x TW tile and walks DZ
// consecutive output planes.  Three input planes live in an LDS ring; each step retires one plane, stores
// the plane that was prefetched into registers during the previous step's MFMAs (the global loads are
// issued before the MFMA block and only waited for after it) and computes one output plane from the three
// resident ones.  Every input plane is staged once per tile column instead of three times, and its L2
// latency is hidden under ~9k cycles of matrix work.
template <int CIN, int NT, int MT, bool PAIR, int DZ>
__global__ void __launch_bounds__(256) conv3d_march_kernel(ConvParams p) {
    using G = KGroup<CIN>;
    constexpr int TW = 16 * MT, SX = PAIR ? 2 : 1;
    constexpr int ROWS = CONV_TH + 2, COLS = PAIR ? 2 * TW + 2 : TW + 2, NWT = PAIR ? 4 : 3;
    // CIN = 32 PAIR (conv0): no per-pixel padding but an XOR swizzle of the 16-B channel chunks by the pixel-pair
    // index, which keeps the ring at 78 KB so that TWO workgroups fit a CU (one computes while the other stages)
    constexpr bool SWZ = PAIR && CIN == 32;
    constexpr int PSTR = SWZ ? CIN : CIN + CONV_PAD, SLAB = ROWS * COLS * PSTR, C4 = CIN / 4;
    constexpr int NEL = ROWS * COLS * C4, NPF = (NEL + 255) / 256;
    constexpr bool SPLIT = MT * NT == 1;  // a single accumulator would serialise on the MFMA's dependent latency
    // Packed weights live in LDS when they fit beside the ring.  vmcnt retires in order, so a weight fragment
    // fetched with global_load inside the tap loop would make every `s_waitcnt` also wait for the (older) plane
    // prefetch and for its own L2 round trip; ds_reads use lgkmcnt and leave the prefetch in flight.
    constexpr int WFLOATS = (PAIR ? 36 : 27) * G::NKG * NT * 64 * G::R;
    constexpr bool WLDS = (size_t)(3 * SLAB + 4 + WFLOATS) * sizeof(float) <= 160 * 1024;
    extern __shared__ __attribute__((aligned(16))) float ring[];  // [3][ROWS][COLS][PSTR] + dummy float4 + weights

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int vox = lane & 15, q = lane >> 4;

    int bx = blockIdx.x;
    const int tw = bx % p.tiles_w; bx /= p.tiles_w;
    const int th = bx % p.tiles_h; bx /= p.tiles_h;
    const int nzc = (p.Do + DZ - 1) / DZ;
    const int zc = bx % nzc;
    const int b = bx / nzc;
    const int r0 = th * CONV_TH, c0 = tw * TW;
    const int z0 = zc * DZ, z1 = min(z0 + DZ, p.Do);
    const int in_r0 = r0 - 1, in_c0 = PAIR ? 2 * c0 - 1 : c0 - 1;

    // this thread's share of a slab: LDS offset and offset inside an input plane (-1: outside the image / unused)
    int loff[NPF], goff[NPF];
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
        const int e = tid + 256 * i;
        const int row = e / (COLS * C4), rem = e - row * (COLS * C4);
        const int col = rem / C4, c4 = rem - col * C4;
        const int gr = in_r0 + row, gc = in_c0 + col;
        const int pc4 = SWZ ? (c4 ^ ((col >> 1) & (C4 - 1))) : c4;
        // float4 units; elements past the slab go to a dummy slot behind the ring, loads outside the image read
        // element 0 and are zeroed by select: every load and store is unconditional (no branches, b128 LDS stores)
        loff[i] = e < NEL ? ((row * COLS + col) * PSTR) / 4 + pc4 : 3 * SLAB / 4;
        goff[i] = (e < NEL && gr >= 0 && gr < p.hi && gc >= 0 && gc < p.wi) ? (gr * p.wi + gc) * C4 + c4 : -1;
    }
    const size_t plane_f4 = (size_t)p.hi * p.wi * C4;
    const float4* __restrict__ xb = reinterpret_cast<const float4*>(p.x) + (size_t)b * p.Di * plane_f4;
    float4* __restrict__ ring4 = reinterpret_cast<float4*>(ring);
    float4 pf[NPF];
    bool pf_ok = false;
    auto load_plane = [&](int plane) {
        const bool ok = plane >= 0 && plane < p.Di;  // block-uniform
        const float4* __restrict__ xp = xb + (size_t)(ok ? plane : 0) * plane_f4;
        pf_ok = ok;
#pragma unroll
        for (int i = 0; i < NPF; ++i) pf[i] = xp[max(goff[i], 0)];  // raw: zeroing happens at store time, so that
    };                                                              // nothing touches pf while the loads are in flight
    auto store_plane = [&](int slot) {
        constexpr int DUMMY = 3 * SLAB / 4;
#pragma unroll
        for (int i = 0; i < NPF; ++i)
            ring4[loff[i] == DUMMY ? DUMMY : slot * (SLAB / 4) + loff[i]] =
                (pf_ok && goff[i] >= 0) ? pf[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    };

    const float* __restrict__ wsrc = p.wpk;
    if constexpr (WLDS) {
        float4* __restrict__ wl4 = ring4 + 3 * SLAB / 4 + 1;
        const float4* __restrict__ wg4 = reinterpret_cast<const float4*>(p.wpk);
        for (int e = tid; e < WFLOATS / 4; e += 256) wl4[e] = wg4[e];
        wsrc = ring + 3 * SLAB + 4;
    }
    // per-lane affine of the 4 output channels this lane owns, fetched once (inside the plane loop these
    // per-lane loads would sit on vmcnt in front of the plane prefetch)
    float esc[NT][4], esh[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ch = (PAIR ? 4 * (q & 1) : n * 16 + q * 4) + k;
            esc[n][k] = ch < p.Cout ? p.scale[ch] : 0.f;
            esh[n][k] = ch < p.Cout ? p.shift[ch] : 0.f;
        }
    load_plane(z0 - 1);
    store_plane((z0 + 2) % 3);
    load_plane(z0);
    store_plane(z0 % 3);
    load_plane(z0 + 1);

    // The float4 stores of plane z are issued at the top of step z+1, BEFORE that step's plane prefetch: vmcnt counts
    // stores too and retires in order, so stores issued after the prefetch (at the end of step z) made the next
    // `s_waitcnt vmcnt(0)` in front of the ring store wait for their write acknowledgement — ~1 us of idle matrix
    // pipe per plane with one wave per SIMD.  Issued first, they have a whole step of MFMAs to complete.
    float4 pend[MT][NT];
    float* pend_p[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) pend_p[m][n] = nullptr;
    auto flush_pending = [&]() {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                if (pend_p[m][n]) *reinterpret_cast<float4*>(pend_p[m][n]) = pend[m][n];
    };

    for (int z = z0; z < z1; ++z) {
        __syncthreads();  // step z-1 no longer reads slot (z+1)%3
        store_plane((z + 1) % 3);
        __syncthreads();
        flush_pending();     // plane z-1's results
        load_plane(z + 2);   // lands during this step's MFMAs (unconditional: behind a branch the prefetch registers
                             // become a phi and the compiler copies — and therefore waits for — them on the spot;
                             // the last step fetches one plane nobody stores)

        f32x4 acc[MT][NT], acc2[SPLIT ? 1 : MT][SPLIT ? 1 : NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (SPLIT) acc2[0][0] = f32x4{0.f, 0.f, 0.f, 0.f};

        // One "step" = one (kd, kh) pair = NWT * NKG fragment groups.  DEEP: the fragments of step s+1 are read from
        // LDS into a second register set before the MFMAs of step s are issued (9 steps fully unrolled, so both
        // sets are statically indexed); a wave then always has ~1k cycles of matrix work in front of any LDS wait.
        constexpr int NGRP = NWT * G::NKG;
        constexpr bool DEEP = G::R == 4 && NGRP * (NT + MT) * G::R * 2 <= 160;
        auto read_step = [&](int step, float (&A)[NGRP][NT][G::R], float (&B)[NGRP][MT][G::R]) {
            const int kd = step / 3, kh = step - kd * 3;
            const float* __restrict__ slab = ring + ((z + kd + 2) % 3) * SLAB;  // plane z + kd - 1
#pragma unroll
            for (int iw = 0; iw < NWT; ++iw) {
                const int tap = PAIR ? step * 4 + iw : step * 3 + iw;
                const float* __restrict__ srow_p = slab + ((wave + kh) * COLS + iw) * PSTR + (SWZ ? 0 : G::R * q);
#pragma unroll
                for (int g = 0; g < G::NKG; ++g) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const float* wp = wsrc + (((tap * G::NKG + g) * NT + n) * 64 + lane) * G::R;
                        if constexpr (G::R == 4) {
                            const float4 t = *reinterpret_cast<const float4*>(wp);
                            A[iw * G::NKG + g][n][0] = t.x; A[iw * G::NKG + g][n][1] = t.y;
                            A[iw * G::NKG + g][n][2] = t.z; A[iw * G::NKG + g][n][3] = t.w;
                        } else {
                            const float2 t = *reinterpret_cast<const float2*>(wp);
                            A[iw * G::NKG + g][n][0] = t.x; A[iw * G::NKG + g][n][1] = t.y;
                        }
                    }
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const float* bp = srow_p + ((m * 16 + vox) * SX) * PSTR + g * G::KG;
                        if constexpr (SWZ) bp += (((g * 4 + q) ^ ((m * 16 + vox + (iw >> 1)) & (C4 - 1))) - g * 4) * 4;
                        if constexpr (G::R == 4) {
                            const float4 t = *reinterpret_cast<const float4*>(bp);
                            B[iw * G::NKG + g][m][0] = t.x; B[iw * G::NKG + g][m][1] = t.y;
                            B[iw * G::NKG + g][m][2] = t.z; B[iw * G::NKG + g][m][3] = t.w;
                        } else {
                            const float2 t = *reinterpret_cast<const float2*>(bp);
                            B[iw * G::NKG + g][m][0] = t.x; B[iw * G::NKG + g][m][1] = t.y;
                        }
                    }
                }
            }
        };
        auto mfma_step = [&](float (&A)[NGRP][NT][G::R], float (&B)[NGRP][MT][G::R]) {
#pragma unroll
            for (int grp = 0; grp < NGRP; ++grp)
#pragma unroll
                for (int j = 0; j < G::R; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            if constexpr (SPLIT) {
                                if (j & 1)
                                    acc2[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[grp][0][j], B[grp][0][j], acc2[0][0], 0, 0, 0);
                                else
                                    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[grp][0][j], B[grp][0][j], acc[0][0], 0, 0, 0);
                            } else {
                                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[grp][n][j], B[grp][m][j], acc[m][n], 0, 0, 0);
                            }
                        }
        };
        if constexpr (DEEP) {
            // one group of the NEXT step is fetched in front of every group of MFMAs of the CURRENT step: with one wave
            // per SIMD nothing else can fill the matrix pipe while this wave issues address arithmetic and ds_reads, so
            // they are spread between the MFMAs (an MFMA occupies the issue port for 8 of its 32 cycles)
            float A0[NGRP][NT][G::R], B0[NGRP][MT][G::R], A1[NGRP][NT][G::R], B1[NGRP][MT][G::R];
            read_step(0, A0, B0);
            auto read_group = [&](int step, int grp, float (&A)[NGRP][NT][G::R], float (&B)[NGRP][MT][G::R]) {
                const int kd = step / 3, kh = step - kd * 3;
                const float* __restrict__ slab = ring + ((z + kd + 2) % 3) * SLAB;
                const int iw = grp / G::NKG, g = grp - iw * G::NKG;
                const int tap = PAIR ? step * 4 + iw : step * 3 + iw;
                const float* __restrict__ srow_p = slab + ((wave + kh) * COLS + iw) * PSTR + (SWZ ? 0 : G::R * q);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float4 t = *reinterpret_cast<const float4*>(wsrc + (((tap * G::NKG + g) * NT + n) * 64 + lane) * G::R);
                    A[grp][n][0] = t.x; A[grp][n][1] = t.y; A[grp][n][2] = t.z; A[grp][n][3] = t.w;
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float* bp = srow_p + ((m * 16 + vox) * SX) * PSTR + g * G::KG;
                    if constexpr (SWZ) bp += (((g * 4 + q) ^ ((m * 16 + vox + (iw >> 1)) & (C4 - 1))) - g * 4) * 4;
                    const float4 t = *reinterpret_cast<const float4*>(bp);
                    B[grp][m][0] = t.x; B[grp][m][1] = t.y; B[grp][m][2] = t.z; B[grp][m][3] = t.w;
                }
            };
            auto mfma_group = [&](int grp, float (&A)[NGRP][NT][G::R], float (&B)[NGRP][MT][G::R]) {
#pragma unroll
                for (int j = 0; j < G::R; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            if constexpr (SPLIT) {
                                if (j & 1)
                                    acc2[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[grp][0][j], B[grp][0][j], acc2[0][0], 0, 0, 0);
                                else
                                    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[grp][0][j], B[grp][0][j], acc[0][0], 0, 0, 0);
                            } else {
                                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[grp][n][j], B[grp][m][j], acc[m][n], 0, 0, 0);
                            }
                        }
            };
            static_assert(G::R == 4, "DEEP path assumes 16-channel k-groups");
#pragma unroll
            for (int step = 0; step < 9; step += 2) {
#pragma unroll
                for (int grp = 0; grp < NGRP; ++grp) {
                    if (step + 1 < 9) read_group(step + 1, grp, A1, B1);
                    mfma_group(grp, A0, B0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (step + 1 < 9) {
#pragma unroll
                    for (int grp = 0; grp < NGRP; ++grp) {
                        if (step + 2 < 9) read_group(step + 2, grp, A0, B0);
                        mfma_group(grp, A1, B1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        } else {
#pragma unroll 1
            for (int step = 0; step < 9; ++step) {
                float A0[NGRP][NT][G::R], B0[NGRP][MT][G::R];
                read_step(step, A0, B0);
                mfma_step(A0, B0);
            }
        }
        if constexpr (SPLIT) acc[0][0] += acc2[0][0];

        // ---- epilogue of plane z (same as conv3d_kernel); full float4s are stored at the top of the next step ----
        const int orow = r0 + wave;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) pend_p[m][n] = nullptr;
        if (orow < p.ho) {
            const size_t row_base = (((size_t)b * p.Do + z) * p.ho + orow) * p.wo;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int gcol = c0 + m * 16 + vox;
                const int ocol = PAIR ? 2 * gcol + (q >> 1) : gcol;
                if (ocol >= p.wo) continue;
                const size_t o = (row_base + ocol) * p.Cout;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int cb = PAIR ? 4 * (q & 1) : n * 16 + q * 4;
                    if (cb >= p.Cout) continue;
                    float r[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int ch = cb + k;
                        float val = 0.f;
                        if (ch < p.Cout) {
                            val = fmaf(acc[m][n][k], esc[n][k], esh[n][k]);
                            if (p.relu) val = fmaxf(val, 0.f);
                            if (p.skip) val += p.skip[o + ch];
                        }
                        r[k] = val;
                    }
                    if (cb + 3 < p.Cout) {
                        pend[m][n] = make_float4(r[0], r[1], r[2], r[3]);
                        pend_p[m][n] = p.y + o + cb;
                    } else {
                        for (int k = 0; k < 4; ++k)
                            if (cb + k < p.Cout) p.y[o + cb + k] = r[k];
                    }
                }
            }
        }
    }
    flush_pending();
}


static long long march_min_blocks() {
    const char* e = exp_env("MVD_K4_MARCH_MIN");
    return e ? atoll(e) : 1024;
}

template <int CIN, int NT, int MT, bool PAIR, int MARCH_DZ = 16>
static int launch_march(const ConvParams& p0, hipStream_t st) {
    ConvParams p = p0;
    constexpr int TW = 16 * MT;
    constexpr int COLS = PAIR ? 2 * TW + 2 : TW + 2;
    constexpr int PSTR = (PAIR && CIN == 32) ? CIN : CIN + CONV_PAD;
    constexpr size_t ring_b = (size_t)3 * (CONV_TH + 2) * COLS * PSTR * sizeof(float) + 16;  // + dummy slot
    constexpr size_t w_b = (size_t)(PAIR ? 36 : 27) * CIN * 16 * NT * sizeof(float);
    constexpr size_t lds = ring_b + w_b <= 160 * 1024 ? ring_b + w_b : ring_b;  // weights in LDS when they fit
    static_assert(ring_b <= 160 * 1024, "ring exceeds LDS");
    const int gw = PAIR ? (p.wo + 1) / 2 : p.wo;
    p.tiles_h = (p.ho + CONV_TH - 1) / CONV_TH;
    p.tiles_w = (gw + TW - 1) / TW;
    const long long nblk = (long long)p.tiles_w * p.tiles_h * ((p.Do + MARCH_DZ - 1) / MARCH_DZ) * p.B;
    if (nblk > 0x7fffffffLL) {
        set_error("conv3d: %lld workgroups exceed the grid limit", nblk);
        return MVD_ERR_INVALID_ARG;
    }
    // too few depth-marching workgroups to fill 256 CUs: shorter plane chunks, then the plane-at-a-time kernel
    if (nblk < march_min_blocks()) {
        if constexpr (MARCH_DZ > 8) return launch_march<CIN, NT, MT, PAIR, 8>(p0, st);
        return -1;
    }
    auto kern = conv3d_march_kernel<CIN, NT, MT, PAIR, MARCH_DZ>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return launch_status("conv3d: LDS attribute");
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, p);
    return launch_status("conv3d_march");
}

// conv0 (32 -> 8, PAIR mode) with the reduction dimension split over TWO waves per output row: a workgroup is 8 waves —
// waves 0-3 take input channels 0-15 of rows 0-3, waves 4-7 channels 16-31 — sharing one ring.  conv3d_march_kernel runs
// this layer with ONE wave per SIMD, so every barrier, ring store, epilogue and every vector-ALU instruction between MFMAs
// (which do not overlap with them on gfx950, tools/micro/mfma_mix.hip) idles the matrix pipe: 69 % busy.  Two waves per SIMD
// fill each other's gaps.  The second half's partial sums (one float4 per lane) go through LDS at the top of the next step,
// where a barrier exists anyway.  Accumulation order: (channels 0-15 over all taps) + (channels 16-31 over all taps).
// Each wave's 36 weight fragments live in registers (144 VGPRs; 233 in all, two waves per SIMD), the ring has FOUR slots.
// Knock-out timings (round 2, 768x1152): the bare MFMA + fragment-read loop 2.01 ms (84 % of the fp32 matrix peak at 75 %
// useful rows), ring write +0.10, the two barriers of the 3-slot form +0.06, epilogue +0.03, prefetch +0.03.
template <int DZ>
__global__ void __launch_bounds__(512) conv0_ksplit_kernel(ConvParams p) {
    constexpr int CIN = 32, TW = 16, SX = 2, ROWS = CONV_TH + 2, COLS = 2 * TW + 2, NWT = 4;
    constexpr int PSTR = CIN, SLAB = ROWS * COLS * PSTR, C4 = CIN / 4, NKG = 2;  // k-groups of 16 channels: one per wave half
    constexpr int NEL = ROWS * COLS * C4, NPF = (NEL + 511) / 512;
    constexpr int DUMMY = 4 * SLAB / 4;  // float4 index of the dummy slot behind the ring
    extern __shared__ __attribute__((aligned(16))) float ring[];  // [4][ROWS][COLS][PSTR] | dummy float4 | partials

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int row = wave & 3, kh_ = wave >> 2;  // output row of the tile, channel half
    const int vox = lane & 15, q = lane >> 4;

    int bx = blockIdx.x;
    const int tw = bx % p.tiles_w; bx /= p.tiles_w;
    const int th = bx % p.tiles_h; bx /= p.tiles_h;
    const int nzc = (p.Do + DZ - 1) / DZ;
    const int zc = bx % nzc;
    const int b = bx / nzc;
    const int r0 = th * CONV_TH, c0 = tw * TW;
    const int z0 = zc * DZ, z1 = min(z0 + DZ, p.Do);
    const int in_r0 = r0 - 1, in_c0 = 2 * c0 - 1;

    int loff[NPF], goff[NPF];
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
        const int e = tid + 512 * i;
        const int rw = e / (COLS * C4), rem = e - rw * (COLS * C4);
        const int col = rem / C4, c4 = rem - col * C4;
        const int gr = in_r0 + rw, gc = in_c0 + col;
        const int pc4 = c4 ^ ((col >> 1) & (C4 - 1));  // XOR swizzle of the 16-B channel chunks by the pixel-pair index
        loff[i] = e < NEL ? ((rw * COLS + col) * PSTR) / 4 + pc4 : DUMMY;
        goff[i] = (e < NEL && gr >= 0 && gr < p.hi && gc >= 0 && gc < p.wi) ? (gr * p.wi + gc) * C4 + c4 : -1;
    }
    const size_t plane_f4 = (size_t)p.hi * p.wi * C4;
    const float4* __restrict__ xb = reinterpret_cast<const float4*>(p.x) + (size_t)b * p.Di * plane_f4;
    float4* __restrict__ ring4 = reinterpret_cast<float4*>(ring);
    float4 pf[NPF];
    bool pf_ok = false;
    auto load_plane = [&](int plane) {
        const bool ok = plane >= 0 && plane < p.Di;  // block-uniform
        const float4* __restrict__ xp = xb + (size_t)(ok ? plane : 0) * plane_f4;
        pf_ok = ok;
#pragma unroll
        for (int i = 0; i < NPF; ++i) pf[i] = xp[max(goff[i], 0)];
    };
    // tiles whose halo lies inside the image (3 of 4 at the headline shape) store the prefetched plane as it is: the
    // zeroing selects are vector-ALU work, which on gfx950 stalls the matrix pipe
    const bool interior = in_r0 >= 0 && in_r0 + ROWS <= p.hi && in_c0 >= 0 && in_c0 + COLS <= p.wi;  // block-uniform
    auto store_plane = [&](int slot) {
        if (interior && pf_ok) {
#pragma unroll
            for (int i = 0; i < NPF; ++i) ring4[loff[i] == DUMMY ? DUMMY : slot * (SLAB / 4) + loff[i]] = pf[i];
        } else {
#pragma unroll
            for (int i = 0; i < NPF; ++i)
                ring4[loff[i] == DUMMY ? DUMMY : slot * (SLAB / 4) + loff[i]] =
                    (pf_ok && goff[i] >= 0) ? pf[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    // LDS: [4 ring slots][dummy float4][partials of the second half, double-buffered by step parity]
    float4* __restrict__ red4 = ring4 + DUMMY + 1;  // [2][4 rows][64 lanes]

    // this wave's 36 weight fragments (its 16 channels x 16 rows x 36 taps) stay in registers: with the 74 KB weight copy out
    // of the LDS the ring can hold a fourth plane, which is what takes the ring write out from between two barriers
    float4 wreg[36];
#pragma unroll
    for (int t = 0; t < 36; ++t) wreg[t] = reinterpret_cast<const float4*>(p.wpk)[(t * NKG + kh_) * 64 + lane];

    float esc[4], esh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        esc[k] = p.scale[4 * (q & 1) + k];
        esh[k] = p.shift[4 * (q & 1) + k];
    }
    // plane P lives in slot P & 3.  Before step z the ring holds z-1, z, z+1; plane z+2 is in flight in registers.
    load_plane(z0 - 1);
    store_plane((z0 - 1) & 3);
    load_plane(z0);
    store_plane(z0 & 3);
    load_plane(z0 + 1);
    store_plane((z0 + 1) & 3);
    load_plane(z0 + 2);

    // first half: its own partial sums of the previous plane, completed and stored at the top of the next step
    f32x4 mine = f32x4{0.f, 0.f, 0.f, 0.f};
    float* mine_p = nullptr;
    const int orow = r0 + row;
    const int ocol = 2 * (c0 + vox) + (q >> 1);
    const bool live = orow < p.ho && ocol < p.wo;
    auto finish_previous = [&](int par) {  // after the barrier that follows the step whose partials sit in red4[par]
        if (kh_ == 0 && mine_p) {
            const float4 o = red4[(par * 4 + row) * 64 + lane];
            float r[4];
            const float a[4] = {mine[0] + o.x, mine[1] + o.y, mine[2] + o.z, mine[3] + o.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r[k] = fmaf(a[k], esc[k], esh[k]);
                if (p.relu) r[k] = fmaxf(r[k], 0.f);
            }
            *reinterpret_cast<float4*>(mine_p) = make_float4(r[0], r[1], r[2], r[3]);
        }
    };

    // ONE barrier per step.  It publishes plane z+1 (written into its slot during step z-1) and the second half's partial
    // sums of step z-1, and it retires every read of slot (z+2)&3 = (z-2)&3 (plane z-2, last read in step z-1), so that plane
    // z+2 can be written into that slot in the MIDDLE of this step's MFMAs: the ring write is no longer a phase of its own
    // between two barriers in which both waves of every SIMD leave the matrix pipe idle (knock-out timings of the 3-slot
    // form: ring write 0.10 ms, second barrier 0.03 ms of 2.26).
    for (int z = z0; z < z1; ++z) {
        __syncthreads();
        finish_previous((z - 1) & 1);  // plane z-1: issued before this step's prefetch (vmcnt retires in order)

        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
        auto read_group = [&](int step, int iw, float (&A)[NWT][4], float (&B)[NWT][4]) {
            const int kd = step / 3, kh = step - kd * 3;
            const float* __restrict__ slab = ring + ((z + kd - 1) & 3) * SLAB;  // plane z + kd - 1
            const int tap = step * 4 + iw;
            const float4 ta = wreg[tap];
            A[iw][0] = ta.x; A[iw][1] = ta.y; A[iw][2] = ta.z; A[iw][3] = ta.w;
            const float* bp = slab + ((row + kh) * COLS + iw) * PSTR + (vox * SX) * PSTR +
                              (((kh_ * 4 + q) ^ ((vox + (iw >> 1)) & (C4 - 1)))) * 4;
            const float4 tb = *reinterpret_cast<const float4*>(bp);
            B[iw][0] = tb.x; B[iw][1] = tb.y; B[iw][2] = tb.z; B[iw][3] = tb.w;
        };
        auto mfma_group = [&](int iw, float (&A)[NWT][4], float (&B)[NWT][4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j & 1) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[iw][j], B[iw][j], acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[iw][j], B[iw][j], acc, 0, 0, 0);
            }
        };
        float A0[NWT][4], B0[NWT][4], A1[NWT][4], B1[NWT][4];
#pragma unroll
        for (int iw = 0; iw < NWT; ++iw) read_group(0, iw, A0, B0);
#pragma unroll
        for (int step = 0; step < 9; step += 2) {
#pragma unroll
            for (int iw = 0; iw < NWT; ++iw) {
                if (step + 1 < 9) read_group(step + 1, iw, A1, B1);
                mfma_group(iw, A0, B0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (step == 4) {  // two thirds of the step's MFMAs are issued: the plane loaded a step ago goes into the ring
                store_plane((z + 2) & 3);
                load_plane(z + 3);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (step + 1 < 9) {
#pragma unroll
                for (int iw = 0; iw < NWT; ++iw) {
                    if (step + 2 < 9) read_group(step + 2, iw, A0, B0);
                    mfma_group(iw, A1, B1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        acc += acc2;
        if (kh_ == 1) {
            red4[((z & 1) * 4 + row) * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        } else {
            mine = acc;
            mine_p = live ? p.y + ((((size_t)b * p.Do + z) * p.ho + orow) * p.wo + ocol) * 8 + 4 * (q & 1) : nullptr;
        }
    }
    __syncthreads();
    finish_previous((z1 - 1) & 1);
}

template <int DZ>
static int launch_conv0_ksplit(const ConvParams& p0, hipStream_t st) {
    ConvParams p = p0;
    constexpr int ROWS = CONV_TH + 2, COLS = 34, SLAB = ROWS * COLS * 32;
    constexpr size_t lds = (size_t)(4 * SLAB + 4 + 2 * 4 * 64 * 4) * sizeof(float);
    static_assert(lds <= 160 * 1024, "conv0 k-split: LDS");
    p.tiles_h = (p.ho + CONV_TH - 1) / CONV_TH;
    p.tiles_w = ((p.wo + 1) / 2 + 15) / 16;
    const long long nblk = (long long)p.tiles_w * p.tiles_h * ((p.Do + DZ - 1) / DZ) * p.B;
    if (nblk > 0x7fffffffLL) {
        set_error("conv3d: %lld workgroups exceed the grid limit", nblk);
        return MVD_ERR_INVALID_ARG;
    }
    if (nblk < march_min_blocks()) {  // too few workgroups to fill 256 CUs: shorter chunks, then the other kernels
        if constexpr (DZ > 16) return launch_conv0_ksplit<16>(p0, st);
        else if constexpr (DZ > 8) return launch_conv0_ksplit<8>(p0, st);
        return -1;
    }
    auto kern = conv0_ksplit_kernel<DZ>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return launch_status("conv3d: LDS attribute");
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), lds, st, p);
    return launch_status("conv0_ksplit");
}

// ConvTranspose3d (k3 s2 p1 op1) with ALL 8 output parity classes in one workgroup: out[2a+p] along each axis
// uses kernel index 1 on input a (p = 0) or indices 0 / 2 on inputs a+1 / a (p = 1), so every one of the 27 taps
// belongs to exactly one class (bit per axis = k != 1) and reads the slab at offset (k == 0).  One slab of
// 2 input planes x (4+1) rows x (TW+1) cols feeds a 2 x 8 x 2*TW output block: the input is staged once instead
// of once per class, and the two w-parities of a row are written back to back (full contiguous rows).
template <int CIN, int NT, int MT>
__global__ void __launch_bounds__(256) deconv3d_all_kernel(ConvParams p) {
    using G = KGroup<CIN>;
    constexpr int TW = 16 * MT, ROWS = CONV_TH + 1, COLS = TW + 1;
    constexpr int PSTR = CIN + CONV_PAD, SLAB = ROWS * COLS * PSTR, C4 = CIN / 4;
    extern __shared__ __attribute__((aligned(16))) float slab[];  // [2 planes][ROWS][COLS][PSTR]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int vox = lane & 15, q = lane >> 4;
    int bx = blockIdx.x;
    const int tw = bx % p.tiles_w; bx /= p.tiles_w;
    const int th = bx % p.tiles_h; bx /= p.tiles_h;
    const int zd = bx % p.Di;
    const int b = bx / p.Di;
    const int r0 = th * CONV_TH, c0 = tw * TW;

    // per-lane affine of this lane's 4 output channels (see conv3d_march_kernel)
    float esc[NT][4], esh[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ch = n * 16 + q * 4 + k;
            esc[n][k] = ch < p.Cout ? p.scale[ch] : 0.f;
            esh[n][k] = ch < p.Cout ? p.shift[ch] : 0.f;
        }

    // weight fragments of the 3 w-taps of one (kd, kh) step in one batch, fetched one step ahead of the MFMAs that use
    // them (a load + wait per step would expose an L2 round trip nine times per workgroup)
    auto load_a = [&](int step, float (&dst)[3][G::NKG][NT][G::R]) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int g = 0; g < G::NKG; ++g)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int tap = step * 3 + kw;
                    const float* wp = p.wpk + ((((size_t)tap * G::NKG + g) * NT + n) * 64 + lane) * G::R;
                    if constexpr (G::R == 4) {
                        const float4 t = *reinterpret_cast<const float4*>(wp);
                        dst[kw][g][n][0] = t.x; dst[kw][g][n][1] = t.y; dst[kw][g][n][2] = t.z; dst[kw][g][n][3] = t.w;
                    } else {
                        const float2 t = *reinterpret_cast<const float2*>(wp);
                        dst[kw][g][n][0] = t.x; dst[kw][g][n][1] = t.y;
                    }
                }
    };
    constexpr bool AHEAD = 2 * 3 * G::NKG * NT * G::R <= 96;  // both batches fit the register file
    float af[3][G::NKG][NT][G::R];
    if constexpr (AHEAD) load_a(0, af);
    // ---- stage input planes zd and zd+1 (zero beyond the volume): all of a thread's loads in flight together ----
    {
        constexpr int NEL = 2 * ROWS * COLS * C4, NPF = (NEL + 255) / 256;
        const float4* __restrict__ x4 = reinterpret_cast<const float4*>(p.x) + (size_t)b * p.Di * p.hi * p.wi * C4;
        float4* __restrict__ s4 = reinterpret_cast<float4*>(slab);
        float4 pf[NPF];
        bool ok[NPF];
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int e = tid + 256 * i;
            const int r = e / (COLS * C4), rem = e - r * (COLS * C4);
            const int col = rem / C4, c4 = rem - col * C4;
            const int pl = r / ROWS, row = r - pl * ROWS;
            const int plane = zd + pl, gr = r0 + row, gc = c0 + col;
            ok[i] = e < NEL && plane < p.Di && gr < p.hi && gc < p.wi;
            pf[i] = x4[ok[i] ? (((size_t)plane * p.hi + gr) * p.wi + gc) * C4 + c4 : 0];
        }
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int e = tid + 256 * i;
            const int pix = e / C4, c4 = e - pix * C4;
            if (e < NEL) s4[(pix * PSTR) / 4 + c4] = ok[i] ? pf[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();

    f32x4 acc[8][MT][NT];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[c][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int step = 0; step < 9; ++step) {
        const int kd = step / 3, kh = step % 3;
        float af_next[3][G::NKG][NT][G::R];
        if constexpr (AHEAD) {
            if (step + 1 < 9) load_a(step + 1, af_next);
        } else {
            load_a(step, af);
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int cls = ((kd != 1) << 2) | ((kh != 1) << 1) | (kw != 1);
            const float* __restrict__ srow_p =
                slab + (kd == 0 ? SLAB : 0) + ((wave + (kh == 0)) * COLS + (kw == 0)) * PSTR + G::R * q;
#pragma unroll
            for (int g = 0; g < G::NKG; ++g) {
                float bf[MT][G::R];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float* bp = srow_p + (m * 16 + vox) * PSTR + g * G::KG;
                    if constexpr (G::R == 4) {
                        const float4 t = *reinterpret_cast<const float4*>(bp);
                        bf[m][0] = t.x; bf[m][1] = t.y; bf[m][2] = t.z; bf[m][3] = t.w;
                    } else {
                        const float2 t = *reinterpret_cast<const float2*>(bp);
                        bf[m][0] = t.x; bf[m][1] = t.y;
                    }
                }
#pragma unroll
                for (int j = 0; j < G::R; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[cls][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kw][g][n][j], bf[m][j], acc[cls][m][n], 0, 0, 0);
            }
        }
        if constexpr (AHEAD) {
            if (step + 1 < 9) {
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int g = 0; g < G::NKG; ++g)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
#pragma unroll
                            for (int j = 0; j < G::R; ++j) af[kw][g][n][j] = af_next[kw][g][n][j];
            }
        }
    }

    // ---- epilogue: 8 classes -> output voxels (2zd+pd, 2row+ph, 2col+pw) ----
    const int arow = r0 + wave;
    if (arow >= p.hi) return;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int pd = c >> 2, ph = (c >> 1) & 1, pw = c & 1;
        const size_t row_base = (((size_t)b * p.Do + 2 * zd + pd) * p.ho + 2 * arow + ph) * p.wo;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int gcol = c0 + m * 16 + vox;
            if (gcol >= p.wi) continue;
            const size_t o = (row_base + 2 * gcol + pw) * p.Cout;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int cb = n * 16 + q * 4;
                if (cb >= p.Cout) continue;
                float4 sk = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.skip) sk = *reinterpret_cast<const float4*>(p.skip + o + cb);  // Cout is a multiple of 4 here
                float r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float val = fmaf(acc[c][m][n][k], esc[n][k], esh[n][k]);
                    if (p.relu) val = fmaxf(val, 0.f);
                    r[k] = val;
                }
                *reinterpret_cast<float4*>(p.y + o + cb) = make_float4(r[0] + sk.x, r[1] + sk.y, r[2] + sk.z, r[3] + sk.w);
            }
        }
    }
}

// ConvTranspose3d with 8 output channels (conv11): the 16 MFMA rows are (output w-parity) x (8 couts), so the two
// w-parity classes of a (d,h)-parity class share one accumulator — 2 MFMA groups per (kd,kh) instead of 3, half the
// accumulators, and the C/D layout (lane q holds parity q>>1, couts 4(q&1)..+3 of input column `vox`) makes every
// store a contiguous 1 KiB: 16 column pairs x 64 B.  Weights packed with mode 3 of pack_weights_kernel.
template <int CIN, int MT>
__global__ void __launch_bounds__(256, 4) deconv3d_pair_kernel(ConvParams p) {
    using G = KGroup<CIN>;
    constexpr int TW = 16 * MT, ROWS = CONV_TH + 1, COLS = TW + 1;
    constexpr int PSTR = CIN + CONV_PAD, SLAB = ROWS * COLS * PSTR, C4 = CIN / 4;
    extern __shared__ __attribute__((aligned(16))) float slab[];  // [2 planes][ROWS][COLS][PSTR]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int vox = lane & 15, q = lane >> 4;
    int bx = tile_index(p);
    const int tw = bx % p.tiles_w; bx /= p.tiles_w;
    const int th = bx % p.tiles_h; bx /= p.tiles_h;
    const int zd = bx % p.Di;
    const int b = bx / p.Di;
    const int r0 = th * CONV_TH, c0 = tw * TW;

    float esc[4], esh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        esc[k] = p.scale[(q & 1) * 4 + k];
        esh[k] = p.shift[(q & 1) * 4 + k];
    }

    // weight fragments of the 2 steps of one (kd, kh), fetched one (kd, kh) ahead of the MFMAs that use them
    auto load_a = [&](int kdh, float (&dst)[2][G::NKG][G::R]) {
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int g = 0; g < G::NKG; ++g) {
                const float* wp = p.wpk + (((size_t)(kdh * 2 + st) * G::NKG + g) * 64 + lane) * G::R;
                if constexpr (G::R == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(wp);
                    dst[st][g][0] = t.x; dst[st][g][1] = t.y; dst[st][g][2] = t.z; dst[st][g][3] = t.w;
                } else {
                    const float2 t = *reinterpret_cast<const float2*>(wp);
                    dst[st][g][0] = t.x; dst[st][g][1] = t.y;
                }
            }
    };
    float af[2][G::NKG][G::R];
    load_a(0, af);

    // ---- stage input planes zd and zd+1 (zero beyond the volume): all of a thread's loads in flight together ----
    {
        constexpr int NEL = 2 * ROWS * COLS * C4, NPF = (NEL + 255) / 256;
        const float4* __restrict__ x4 = reinterpret_cast<const float4*>(p.x) + (size_t)b * p.Di * p.hi * p.wi * C4;
        float4* __restrict__ s4 = reinterpret_cast<float4*>(slab);
        float4 pf[NPF];
        bool ok[NPF];
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int e = tid + 256 * i;
            const int r = e / (COLS * C4), rem = e - r * (COLS * C4);
            const int col = rem / C4, c4 = rem - col * C4;
            const int pl = r / ROWS, row = r - pl * ROWS;
            const int plane = zd + pl, gr = r0 + row, gc = c0 + col;
            ok[i] = e < NEL && plane < p.Di && gr < p.hi && gc < p.wi;
            pf[i] = x4[ok[i] ? (((size_t)plane * p.hi + gr) * p.wi + gc) * C4 + c4 : 0];
        }
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int e = tid + 256 * i;
            const int pix = e / C4, c4 = e - pix * C4;
            if (e < NEL) s4[(pix * PSTR) / 4 + c4] = ok[i] ? pf[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();

    f32x4 acc[4][MT];  // (d,h)-parity class x column tile
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int kdh = 0; kdh < 9; ++kdh) {
        const int kd = kdh / 3, kh = kdh % 3;
        float af_next[2][G::NKG][G::R];
        if (kdh + 1 < 9) load_a(kdh + 1, af_next);
        const int cls = ((kd != 1) << 1) | (kh != 1);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const float* __restrict__ srow_p = slab + (kd == 0 ? SLAB : 0) + ((wave + (kh == 0)) * COLS + st) * PSTR + G::R * q;
#pragma unroll
            for (int g = 0; g < G::NKG; ++g) {
                float bf[MT][G::R];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float* bp = srow_p + (m * 16 + vox) * PSTR + g * G::KG;
                    if constexpr (G::R == 4) {
                        const float4 t = *reinterpret_cast<const float4*>(bp);
                        bf[m][0] = t.x; bf[m][1] = t.y; bf[m][2] = t.z; bf[m][3] = t.w;
                    } else {
                        const float2 t = *reinterpret_cast<const float2*>(bp);
                        bf[m][0] = t.x; bf[m][1] = t.y;
                    }
                }
#pragma unroll
                for (int j = 0; j < G::R; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[cls][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[st][g][j], bf[m][j], acc[cls][m], 0, 0, 0);
            }
        }
        if (kdh + 1 < 9) {
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int g = 0; g < G::NKG; ++g)
#pragma unroll
                    for (int j = 0; j < G::R; ++j) af[st][g][j] = af_next[st][g][j];
        }
    }

    // ---- epilogue: lane (q, vox) owns couts 4(q&1)..+3 of output voxel (2zd+pd, 2row+ph, 2(c0+16m+vox) + (q>>1)) ----
    // Branch-free through buffer instructions (a column beyond the row, a missing skip tensor = an out-of-range offset / an
    // empty descriptor): the skip reads of all four parity classes can then be in flight together instead of one exposed
    // round trip per class (as many as fit the 128 registers of four waves per SIMD: 0.300 -> 0.270 ms for conv11; letting the
    // kernel grow to 188 registers and two waves per SIMD gave the gain back).
    const int arow = r0 + wave;
    if (arow >= p.hi) return;  // wave-uniform
    constexpr unsigned OOB = 0x80000000u;
    unsigned coff[MT];  // byte offset of this lane's float4 inside an output ROW
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int gcol = c0 + m * 16 + vox;
        coff[m] = gcol < p.wi ? (unsigned)(((size_t)2 * gcol * 8 + q * 4) * 4) : OOB;
    }
    const int row_bytes = p.wo * 8 * 4;
    float4 sk[4][MT];
    __amdgpu_buffer_rsrc_t yrs[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int pd = c >> 1, ph = c & 1;
        const size_t row_base = (((size_t)b * p.Do + 2 * zd + pd) * p.ho + 2 * arow + ph) * p.wo * 8;
        yrs[c] = __builtin_amdgcn_make_buffer_rsrc(p.y + row_base, 0, row_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.skip ? p.skip + row_base : p.y), 0, p.skip ? row_bytes : 0, 0x00020000);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const u32x4c t = __builtin_amdgcn_raw_buffer_load_b128(srs, coff[m], 0, 0);
            sk[c][m] = make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]));
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r[k] = fmaf(acc[c][m][k], esc[k], esh[k]);
                if (p.relu) r[k] = fmaxf(r[k], 0.f);
            }
            __builtin_amdgcn_raw_buffer_store_b128(u32x4c{__float_as_uint(r[0] + sk[c][m].x), __float_as_uint(r[1] + sk[c][m].y),
                                                          __float_as_uint(r[2] + sk[c][m].z), __float_as_uint(r[3] + sk[c][m].w)},
                                                   yrs[c], coff[m], 0, 0);
        }
}

template <int CIN, int MT>
static int launch_deconv_pair(const ConvParams& p0, hipStream_t st) {
    ConvParams p = p0;
    constexpr size_t lds = (size_t)2 * (CONV_TH + 1) * (16 * MT + 1) * (CIN + CONV_PAD) * sizeof(float);
    static_assert(lds <= 160 * 1024, "slab exceeds LDS");
    p.tiles_h = (p.hi + CONV_TH - 1) / CONV_TH;
    p.tiles_w = (p.wi + 16 * MT - 1) / (16 * MT);
    const long long nblk = (long long)p.tiles_w * p.tiles_h * p.Di * p.B;
    if (nblk > 0x7fffffffLL) {
        set_error("conv3d: %lld workgroups exceed the grid limit", nblk);
        return MVD_ERR_INVALID_ARG;
    }
    p.per_xcd = xcd_run(nblk, 1);
    auto kern = deconv3d_pair_kernel<CIN, MT>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return launch_status("conv3d: LDS attribute");
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, p);
    return launch_status("deconv3d_pair");
}

template <int CIN, int NT, int MT>
static int launch_deconv_all(const ConvParams& p0, hipStream_t st) {
    ConvParams p = p0;
    constexpr size_t lds = (size_t)2 * (CONV_TH + 1) * (16 * MT + 1) * (CIN + CONV_PAD) * sizeof(float);
    static_assert(lds <= 160 * 1024, "slab exceeds LDS");
    p.tiles_h = (p.hi + CONV_TH - 1) / CONV_TH;
    p.tiles_w = (p.wi + 16 * MT - 1) / (16 * MT);
    const long long nblk = (long long)p.tiles_w * p.tiles_h * p.Di * p.B;
    if (nblk > 0x7fffffffLL) {
        set_error("conv3d: %lld workgroups exceed the grid limit", nblk);
        return MVD_ERR_INVALID_ARG;
    }
    auto kern = deconv3d_all_kernel<CIN, NT, MT>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return launch_status("conv3d: LDS attribute");
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, p);
    return launch_status("deconv3d_all");
}

// `prob`: 3x3x3, 8 -> 1 channels, stride 1 (mvsnet_components.py:109).  One GEMM row of 16 would be used on
// the matrix cores, so this layer runs on the vector ALU: one lane per output voxel, the three input planes
// of a 4 x 64 tile resident in LDS (48-B pixels: conflict-free ds_read_b128 across consecutive columns), the
// 216 weights through the scalar cache.  packed weights here are [tap 27][cin 8].
constexpr int C8_DZ = 16;  // output planes a workgroup of the 8 -> 1 kernel walks
constexpr int C8_TW = 62;  // output columns per tile row: 64 slab columns (one per lane) minus the two halo columns
__global__ void __launch_bounds__(256, 4) conv3d_c8_to_1_kernel(ConvParams p) {
    // depth-marching like conv3d_march_kernel: 3 input planes of the 4 x 64 tile in an LDS ring, the next plane
    // prefetched into registers while the current output plane is computed
    // Lane l owns slab COLUMN l (input pixel c0 - 1 + l) and, for l = 1..62, output column c0 + l - 1.  Instead of reading
    // its three kw-neighbours' pixels from LDS, a lane forms the three partial sums Q_kw = sum_{kd,kh,c} W[kd,kh,kw,c] *
    // x[own pixel] and the output is Q_0[l-1] + Q_1[l] + Q_2[l+1]: one pixel read per (kd,kh) instead of three (the LDS
    // pipe bounded this kernel) at the price of two lane shifts per output voxel and 62-wide tiles.
    // 32-byte LDS pixels (no padding: two-way bank conflicts on the few remaining reads) keep the ring at 37 KB, four
    // workgroups per CU
    constexpr int CIN = 8, TW = C8_TW, ROWS = CONV_TH + 2, COLS = TW + 2, PSTR = CIN, SLAB = ROWS * COLS * PSTR;
    static_assert(COLS == 64, "one slab column per lane");
    constexpr int NEL = ROWS * COLS * 2, NPF = (NEL + 255) / 256, DZ = C8_DZ;
    __shared__ __attribute__((aligned(16))) float ring[3 * SLAB + 4];  // 37 KB + dummy slot
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx = tile_index(p);
    const int tw = bx % p.tiles_w; bx /= p.tiles_w;
    const int th = bx % p.tiles_h; bx /= p.tiles_h;
    const int nzc = (p.Do + DZ - 1) / DZ;
    const int zc = bx % nzc;
    const int b = bx / nzc;
    const int r0 = th * CONV_TH, c0 = tw * TW;
    const int z0 = zc * DZ, z1 = min(z0 + DZ, p.Do);

    int loff[NPF], goff[NPF];
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
        const int e = tid + 256 * i;
        const int row = e / (COLS * 2), rem = e - row * (COLS * 2);
        const int col = rem >> 1, c4 = rem & 1;
        const int gr = r0 - 1 + row, gc = c0 - 1 + col;
        constexpr int DUMMY = (3 * SLAB) / 4;
        loff[i] = e < NEL ? ((row * COLS + col) * PSTR) / 4 + c4 : DUMMY;  // float4 units (see conv3d_march_kernel)
        goff[i] = (e < NEL && gr >= 0 && gr < p.hi && gc >= 0 && gc < p.wi) ? (gr * p.wi + gc) * 2 + c4 : -1;
    }
    const size_t plane_f4 = (size_t)p.hi * p.wi * 2;
    const float4* __restrict__ xb = reinterpret_cast<const float4*>(p.x) + (size_t)b * p.Di * plane_f4;
    float4* __restrict__ ring4 = reinterpret_cast<float4*>(ring);
    float4 pf[NPF];
    bool pf_ok = false;
    auto load_plane = [&](int plane) {
        const bool ok = plane >= 0 && plane < p.Di;
        const float4* __restrict__ xp = xb + (size_t)(ok ? plane : 0) * plane_f4;
        pf_ok = ok;
#pragma unroll
        for (int i = 0; i < NPF; ++i) pf[i] = xp[max(goff[i], 0)];  // raw: zeroing happens at store time, so that
    };                                                              // nothing touches pf while the loads are in flight
    auto store_plane = [&](int slot) {
        constexpr int DUMMY = (3 * SLAB) / 4;
#pragma unroll
        for (int i = 0; i < NPF; ++i)
            ring4[loff[i] == DUMMY ? DUMMY : slot * (SLAB / 4) + loff[i]] =
                (pf_ok && goff[i] >= 0) ? pf[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    load_plane(z0 - 1);
    store_plane((z0 + 2) % 3);
    load_plane(z0);
    store_plane(z0 % 3);
    load_plane(z0 + 1);

    const int orow = r0 + wave, ocol = c0 + lane - 1;
    const bool live = lane >= 1 && lane <= TW && orow < p.ho && ocol < p.wo;
    const float sc = p.scale[0], sh = p.shift[0];
    for (int z = z0; z < z1; ++z) {
        __syncthreads();
        store_plane((z + 1) % 3);
        __syncthreads();
        if (z + 1 < z1) load_plane(z + 2);
        // per kw two chains of packed FMAs over the channel pairs (0,1),(4,5) and (2,3),(6,7) as they sit in the 128-bit
        // LDS reads (written with vector types: left to itself the vectoriser pairs channels across registers and
        // spends three moves per packed FMA)
        f32x2 qa[3], qb[3];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) { qa[kw] = f32x2{0.f, 0.f}; qb[kw] = f32x2{0.f, 0.f}; }
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const float* __restrict__ slab = ring + ((z + kd + 2) % 3) * SLAB;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const f32x4* sp = reinterpret_cast<const f32x4*>(slab + ((wave + kh) * COLS + lane) * PSTR);
                const f32x4 a = sp[0], c = sp[1];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    // constant address space: the loads are invariant for the compiler, so the (uniform) address goes
                    // through the scalar cache and the weights arrive as SGPR-pair operands of the packed FMAs
                    typedef const __attribute__((address_space(4))) f32x4* cf4;
                    const cf4 wp = (cf4)(unsigned long long)(p.wpk + ((kd * 3 + kh) * 3 + kw) * 8);
                    const f32x4 w0 = wp[0], w1 = wp[1];
                    qa[kw] = __builtin_elementwise_fma(a.xy, w0.xy, qa[kw]);
                    qb[kw] = __builtin_elementwise_fma(a.zw, w0.zw, qb[kw]);
                    qa[kw] = __builtin_elementwise_fma(c.xy, w1.xy, qa[kw]);
                    qb[kw] = __builtin_elementwise_fma(c.zw, w1.zw, qb[kw]);
                }
            }
        }
        float q[3];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) q[kw] = (qa[kw].x + qa[kw].y) + (qb[kw].x + qb[kw].y);
        // output column o = pixel l takes kw = 0 from pixel l-1, kw = 1 from itself, kw = 2 from pixel l+1
        const float acc = (__shfl_up(q[0], 1) + q[1]) + __shfl_down(q[2], 1);
        if (live) {
            const size_t o = (((size_t)b * p.Do + z) * p.ho + orow) * p.wo + ocol;
            float val = fmaf(acc, sc, sh);
            if (p.relu) val = fmaxf(val, 0.f);
            if (p.skip) val += p.skip[o];
            p.y[o] = val;
        }
    }
}

__global__ void pack_c8_to_1_kernel(const float* __restrict__ w, float* __restrict__ packed) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;  // packed [tap][cin] <- w (1, 8, 27)
    if (e < 216) packed[e] = w[(e % 8) * 27 + e / 8];
}

template <int CIN, int NT, int MT, int MODE>
static int launch_conv(const ConvParams& p0, hipStream_t st) {
    ConvParams p = p0;
    using T = TileGeom<MODE, MT>;
    constexpr size_t lds = (size_t)T::ROWS * T::COLS * (CIN + CONV_PAD) * sizeof(float) + 16;  // + dummy float4
    static_assert(lds <= 160 * 1024, "slab exceeds LDS");
    // tiles over the GEMM-column grid: output voxels (conv), voxel pairs (PAIR) or input voxels (deconv)
    const int gh = T::DECONV ? p.hi : p.ho;
    const int gw = T::DECONV ? p.wi : T::PAIR ? (p.wo + 1) / 2 : p.wo;
    p.tiles_h = (gh + CONV_TH - 1) / CONV_TH;
    p.tiles_w = (gw + T::TW - 1) / T::TW;
    const long long nz = T::DECONV ? (long long)p.Di * 4 : p.Do;
    const long long nblk = (long long)p.tiles_w * p.tiles_h * nz * p.B;
    if (nblk > 0x7fffffffLL) {
        set_error("conv3d: %lld workgroups exceed the grid limit", nblk);
        return MVD_ERR_INVALID_ARG;
    }
    auto kern = conv3d_kernel<CIN, NT, MT, MODE>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return launch_status("conv3d: LDS attribute");
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, p);
    return launch_status("conv3d");
}

static int launch_c8_to_1(const ConvParams& p0, hipStream_t st) {
    ConvParams p = p0;
    p.tiles_h = (p.ho + CONV_TH - 1) / CONV_TH;
    p.tiles_w = (p.wo + C8_TW - 1) / C8_TW;
    const long long nblk = (long long)p.tiles_w * p.tiles_h * ((p.Do + C8_DZ - 1) / C8_DZ) * p.B;
    if (nblk > 0x7fffffffLL) {
        set_error("conv3d: %lld workgroups exceed the grid limit", nblk);
        return MVD_ERR_INVALID_ARG;
    }
    p.per_xcd = xcd_run(nblk, 0);
    hipLaunchKernelGGL(conv3d_c8_to_1_kernel, dim3((unsigned)nblk), dim3(256), 0, st, p);
    return launch_status("conv3d_c8_to_1");
}

// 16-column tiles per wave row, lo..hi: the choice that pads a row of `cols` GEMM columns least (ties: the widest)
static int best_mt(int cols, int lo, int hi) {
    int best = hi, waste = 1 << 30;
    for (int mt = hi; mt >= lo; --mt) {
        const int tw = 16 * mt, pad = (cols + tw - 1) / tw * tw - cols;
        if (pad < waste) { waste = pad; best = mt; }
    }
    return best;
}

// the plane-at-a-time kernel with the tile width (MTMAX or one step narrower) that pads the row least
template <int CIN, int NT, int MTMAX, int MODE>
static int launch_conv_best(const ConvParams& p, hipStream_t st) {
    const int gw = MODE == MVD_DECONV3D_STRIDE2 ? p.wi : p.wo;
    if constexpr (MTMAX >= 4) {
        if (best_mt(gw, 3, 4) == 3) return launch_conv<CIN, NT, 3, MODE>(p, st);
    } else if constexpr (MTMAX == 2) {
        if (best_mt(gw, 1, 2) == 1) return launch_conv<CIN, NT, 1, MODE>(p, st);
    }
    return launch_conv<CIN, NT, MTMAX, MODE>(p, st);
}

template <int CIN, int NT, int MTMAX>
static int launch_march_best(const ConvParams& p, hipStream_t st) {
    if constexpr (MTMAX >= 4) {
#ifdef MVD_EXPERIMENTS
        if (const char* e = exp_env("MVD_K4_MARCH_MT")) {  // forced tile width (narrower tiles = more workgroups per CU): conv2
            if (atoi(e) == 1) return launch_march<CIN, NT, 1, false>(p, st);  // 0.267 (3) / 0.271 (2) / 0.247 ms (1)
            if (atoi(e) == 2) return launch_march<CIN, NT, 2, false>(p, st);
        }
#endif
        if (best_mt(p.wo, 3, 4) == 3) return launch_march<CIN, NT, 3, false>(p, st);
    }
    return launch_march<CIN, NT, MTMAX, false>(p, st);
}

template <int CIN, int MODE>
static int dispatch_cout(const ConvParams& p, hipStream_t st) {
    // MT (16-column tiles per wave) chosen so that the slab fits LDS and wide rows get long tiles
    constexpr int MT = MODE == MVD_CONV3D_STRIDE2 ? (CIN >= 32 ? 2 : 4) : 4;
    if constexpr (MODE == MVD_CONV3D_STRIDE1) {
        const bool old = exp_env("MVD_K4_NOMARCH") != nullptr;  // experiments: the plane-at-a-time kernels
        if (p.Cout == 8) {
            if constexpr (CIN == 32)
                if (!old && !p.skip && !exp_env("MVD_K4_NOKSPLIT")) {  // conv0: two waves per row, split over the input channels
                    int rc = -1;
                    const char* dz = exp_env("MVD_K4_CONV0_DZ");  // experiments library: planes per workgroup march
                    if (dz && atoi(dz) == 32) rc = launch_conv0_ksplit<32>(p, st);
                    else if (dz && atoi(dz) == 64) rc = launch_conv0_ksplit<64>(p, st);
                    else rc = launch_conv0_ksplit<16>(p, st);
                    if (rc >= 0) return rc;
                }
            if constexpr (CIN <= 32)
                if (!old) {
                    const int rc = launch_march<CIN, 1, (CIN >= 32 ? 1 : 2), true>(p, st);
                    if (rc >= 0) return rc;
                }
            return launch_conv<CIN, 1, (CIN >= 64 ? 1 : 2), MVD_CONV3D_S1_PAIR>(p, st);
        }
        if (p.Cout == 1 && CIN == 8) return launch_c8_to_1(p, st);
        if (!old && CIN >= 16) {
            constexpr int MM = CIN == 16 ? 4 : CIN == 32 ? 2 : 1;
            int rc = -1;
            switch ((p.Cout + 15) / 16) {
                case 1: rc = launch_march_best<CIN, 1, MM>(p, st); break;
                case 2: rc = launch_march_best<CIN, 2, MM>(p, st); break;
                case 4: rc = launch_march_best<CIN, 4, MM>(p, st); break;
            }
            if (rc >= 0) return rc;
        }
    }
    const int nt = (p.Cout + 15) / 16;
    if constexpr (MODE == MVD_DECONV3D_STRIDE2) {
        // all 8 parity classes per workgroup (Cout a multiple of 4); MVD_K4_DECONV_CLASSES=1 keeps the per-class kernel
        if (p.Cout == 8) {  // its weights are packed in pair form
            if constexpr (CIN >= 64) {
                return launch_deconv_pair<CIN, 2>(p, st);
            } else {
                switch (best_mt(p.wi, 2, 4)) {  // fewest padded columns
                    case 2: return launch_deconv_pair<CIN, 2>(p, st);
                    case 3: return launch_deconv_pair<CIN, 3>(p, st);
                    default: return launch_deconv_pair<CIN, 4>(p, st);
                }
            }
        }
        if (p.Cout % 4 == 0 && nt <= 2 && !exp_env("MVD_K4_DECONV_CLASSES")) {
            if (nt == 1) return launch_deconv_all<CIN, 1, 2>(p, st);
            return launch_deconv_all<CIN, 2, (CIN >= 64 ? 1 : 2)>(p, st);
        }
    }
    switch (nt) {
        case 1: return launch_conv_best<CIN, 1, MT, MODE>(p, st);
        case 2: return launch_conv_best<CIN, 2, MT, MODE>(p, st);
        case 4: return launch_conv_best<CIN, 4, (MODE == MVD_DECONV3D_STRIDE2 ? 2 : MT), MODE>(p, st);
    }
    set_error("conv3d: Cout=%d unsupported", p.Cout);
    return MVD_ERR_INVALID_ARG;
}

template <int CIN>
static int dispatch_mode(const ConvParams& p, int mode, hipStream_t st) {
    switch (mode) {
        case MVD_CONV3D_STRIDE1: return dispatch_cout<CIN, MVD_CONV3D_STRIDE1>(p, st);
        case MVD_CONV3D_STRIDE2: return dispatch_cout<CIN, MVD_CONV3D_STRIDE2>(p, st);
        case MVD_DECONV3D_STRIDE2: return dispatch_cout<CIN, MVD_DECONV3D_STRIDE2>(p, st);
    }
    set_error("conv3d: mode=%d unknown", mode);
    return MVD_ERR_INVALID_ARG;
}

static bool cin_ok(int c) { return c == 8 || c == 16 || c == 32 || c == 64; }
static bool cout_ok(int c) { return c == 1 || c == 8 || c == 16 || c == 32 || c == 64; }

}  // namespace mvd

extern "C" {

size_t mvd_conv3d_packed_weight_floats(int Cin, int Cout) {
    if (!mvd::cin_ok(Cin) || !mvd::cout_ok(Cout)) return 0;
    // 36 taps when a stride-1 layer with 8 output channels is packed for PAIR mode (the larger of the forms a
    // (Cin, Cout) pair can take: the transposed pair form has 18)
    return (size_t)(Cout == 8 ? 36 : 27) * Cin * 16 * ((Cout + 15) / 16);
}

int mvd_pack_conv3d_weights_f32(const float* w, int Cin, int Cout, int mode, float* packed, mvd_stream_t stream) {
    MVD_REQUIRE(w && packed, "pack_conv3d_weights: NULL argument");
    MVD_REQUIRE(mvd::cin_ok(Cin) && mvd::cout_ok(Cout), "pack_conv3d_weights: Cin=%d/Cout=%d unsupported", Cin, Cout);
    MVD_REQUIRE(mode >= 0 && mode <= 2, "pack_conv3d_weights: mode=%d unknown", mode);
    const int NT = (Cout + 15) / 16;
    // 1: ConvTranspose3d weight layout; 2: PAIR-mode packing (stride-1 conv with 8 output channels)
    // 3: transposed conv with 8 output channels, w-parity pair form (deconv3d_pair_kernel)
    const int tr = mode == MVD_DECONV3D_STRIDE2 ? (Cout == 8 ? 3 : 1) : (mode == MVD_CONV3D_STRIDE1 && Cout == 8) ? 2 : 0;
    hipStream_t st = (hipStream_t)stream;
    if (mode == MVD_CONV3D_STRIDE1 && Cout == 1 && Cin == 8) {  // the vector-ALU `prob` kernel takes [tap][cin]
        hipLaunchKernelGGL(mvd::pack_c8_to_1_kernel, dim3(1), dim3(256), 0, st, w, packed);
        return mvd::launch_status("pack_conv3d_weights");
    }
    const unsigned nb = (unsigned)((mvd_conv3d_packed_weight_floats(Cin, Cout) + 255) / 256);
    switch (Cin) {
        case 8: hipLaunchKernelGGL(mvd::pack_weights_kernel<8>, dim3(nb), dim3(256), 0, st, w, packed, Cout, NT, tr); break;
        case 16: hipLaunchKernelGGL(mvd::pack_weights_kernel<16>, dim3(nb), dim3(256), 0, st, w, packed, Cout, NT, tr); break;
        case 32: hipLaunchKernelGGL(mvd::pack_weights_kernel<32>, dim3(nb), dim3(256), 0, st, w, packed, Cout, NT, tr); break;
        case 64: hipLaunchKernelGGL(mvd::pack_weights_kernel<64>, dim3(nb), dim3(256), 0, st, w, packed, Cout, NT, tr); break;
    }
    return mvd::launch_status("pack_conv3d_weights");
}

static int conv3d_entry(const float* x, const float* packed_w, const float* scale, const float* shift, const float* skip, float* y,
                        float* absmax_out, int B, int Di, int hi, int wi, int Cin, int Cout, int mode, int relu, mvd_stream_t stream);

int mvd_conv3d_bn_relu_f32(const float* x, const float* packed_w, const float* scale, const float* shift,
                           const float* skip, float* y, int B, int Di, int hi, int wi, int Cin, int Cout, int mode,
                           int relu, mvd_stream_t stream) {
    return conv3d_entry(x, packed_w, scale, shift, skip, y, nullptr, B, Di, hi, wi, Cin, Cout, mode, relu, stream);
}

int mvd_conv3d_bn_relu_absmax_f32(const float* x, const float* packed_w, const float* scale, const float* shift,
                                  const float* skip, float* y, float* absmax_out, int B, int Di, int hi, int wi, int Cin,
                                  int Cout, int mode, int relu, mvd_stream_t stream) {
    MVD_REQUIRE(absmax_out, "conv3d_absmax: NULL argument");
    return conv3d_entry(x, packed_w, scale, shift, skip, y, absmax_out, B, Di, hi, wi, Cin, Cout, mode, relu, stream);
}

static int conv3d_entry(const float* x, const float* packed_w, const float* scale, const float* shift, const float* skip, float* y,
                        float* absmax_out, int B, int Di, int hi, int wi, int Cin, int Cout, int mode, int relu, mvd_stream_t stream) {
    MVD_REQUIRE(x && packed_w && scale && shift && y, "conv3d: NULL argument");
    MVD_REQUIRE(B > 0 && Di > 0 && hi > 0 && wi > 0, "conv3d: non-positive dimension");
    MVD_REQUIRE(mvd::cin_ok(Cin) && mvd::cout_ok(Cout), "conv3d: Cin=%d/Cout=%d unsupported", Cin, Cout);
    mvd::ConvParams p{};
    p.x = x; p.wpk = packed_w; p.scale = scale; p.shift = shift; p.skip = skip; p.y = y;
    p.B = B; p.Di = Di; p.hi = hi; p.wi = wi; p.Cout = Cout; p.relu = relu;
    if (mode == MVD_CONV3D_STRIDE1) {
        p.Do = Di; p.ho = hi; p.wo = wi;
    } else if (mode == MVD_CONV3D_STRIDE2) {
        MVD_REQUIRE(Di % 2 == 0 && hi % 2 == 0 && wi % 2 == 0, "conv3d stride 2: odd input dims %dx%dx%d", Di, hi, wi);
        p.Do = Di / 2; p.ho = hi / 2; p.wo = wi / 2;
    } else if (mode == MVD_DECONV3D_STRIDE2) {
        p.Do = Di * 2; p.ho = hi * 2; p.wo = wi * 2;
    } else {
        mvd::set_error("conv3d: mode=%d unknown", mode);
        return MVD_ERR_INVALID_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    // max |y|: a by-product of the store epilogue where the layer runs on conv3d_kernel (every stride-2 layer: the regulariser's
    // conv1 and conv3, whose outputs the split-operand conv2 and conv4 consume), a pass over y after any other kernel
    const bool fused = absmax_out && mode == MVD_CONV3D_STRIDE2;
    if (fused) {
        if (hipMemsetAsync(absmax_out, 0, sizeof(float), st) != hipSuccess) return mvd::launch_status("conv3d_absmax: memset");
        p.absmax = absmax_out;
    }
    int rc = MVD_ERR_INVALID_ARG;
    switch (Cin) {
        case 8: rc = mvd::dispatch_mode<8>(p, mode, st); break;
        case 16: rc = mvd::dispatch_mode<16>(p, mode, st); break;
        case 32: rc = mvd::dispatch_mode<32>(p, mode, st); break;
        case 64: rc = mvd::dispatch_mode<64>(p, mode, st); break;
    }
    if (rc == MVD_OK && absmax_out && !fused)
        rc = mvd::absmax_launch(y, (long long)B * p.Do * p.ho * p.wo * Cout, absmax_out, st);
    return rc;
}
}
