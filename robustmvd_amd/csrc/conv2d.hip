// K6 — one layer of MVSNet's FeatureNet (rmvd/models/blocks/mvsnet_components.py:8-22 ConvBnReLU, :44-66
// FeatureNet) as an implicit GEMM on the gfx950 fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 fmaf
// chains): Conv2d k x k (k = 3 stride 1, k = 5 stride 2; padding k/2) with the eval-mode BatchNorm folded into a
// per-channel scale/shift and the ReLU fused into the epilogue.  The reference runs conv, batch-norm and ReLU as
// three passes over the activation; here a layer reads its input once and writes its output once.
//
// GEMM view per kernel tap, as in K4 (conv3d.hip): D[cout, pixel] += W_tap[cout, cin] * X[cin, pixel*stride + tap].
//   A operand = weights (M = 16 couts), pre-packed in fragment order, read from L1/L2 one kernel row ahead;
//   B operand = activations (N = 16 consecutive output pixels of one row) from an LDS slab holding the tile's
//               input rows and halo, channel-last;
//   K         = input channels in groups of KG = min(Cin, 16); one ds_read of R = KG/4 consecutive channels per
//               lane feeds R MFMAs.
// Workgroup = 4 waves; wave r owns RPW consecutive output rows of a (4*RPW) x (16*MT) pixel tile.  Activations
// are channel-last (B,h,w,C) between layers; the first layer reads the reference's (B,3,H,W) image directly and
// the last can write the reference's (B,C,h,w) or straight into K3's zero-bordered staging layout.
#include "mvd_common.h"

namespace mvd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Conv2dParams {
    const float* x;
    const float* wpk;
    const float* scale;
    const float* shift;
    float* y;
    float* absmax;  // optional (device, one float; the caller zeroes it): raised to max |y| over the finite outputs
    int B, hi, wi, ho, wo, Cout, relu;
    int tiles_w, tiles_h, tiles_per_xcd;
    long long osb, oorg;  // output batch stride and origin offset (floats)
    int osr, osc, osch;   // row, column and channel strides inside one image (floats; an image holds < 2^31)
};

template <int CIN>
struct KGroup2 {
    static constexpr int KG = CIN >= 16 ? 16 : CIN;  // channels per k-group
    static constexpr int R = KG / 4;                 // consecutive channels per lane = MFMAs per group
    static constexpr int NKG = CIN / KG;
};

// packed weight index: [tap ks*ks][k-group][n-tile][lane 64][R]; CIN is the padded channel count (3 -> 4)
__global__ void pack_weights2d_kernel(const float* __restrict__ w, float* __restrict__ packed, int cin_real, int CIN,
                                      int Cout, int ks) {
    const int KG = CIN >= 16 ? 16 : CIN, R = KG / 4, NKG = CIN / KG, NT = (Cout + 15) / 16;
    const size_t total = (size_t)ks * ks * NKG * NT * 64 * R;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t r = e;
        const int j = r % R; r /= R;
        const int lane = r % 64; r /= 64;
        const int nt = r % NT; r /= NT;
        const int g = r % NKG; r /= NKG;
        const int tap = (int)r;  // kh * ks + kw
        const int row = nt * 16 + (lane & 15);
        const int cin = g * KG + R * (lane >> 4) + j;
        packed[e] = (row < Cout && cin < cin_real) ? w[((size_t)row * cin_real + cin) * ks * ks + tap] : 0.f;  // (Cout,Cin,k,k)
    }
}

template <int CIN, int NT, int MT, int RPW, int KS, int ST, bool PLANAR>
struct Tile2 {
    static constexpr int TH = 4 * RPW, TW = 16 * MT;
    static constexpr int ROWS = (TH - 1) * ST + KS, COLS = (TW - 1) * ST + KS;
    static constexpr int PSTR = CIN + 4;  // floats per LDS pixel: 16-B aligned, spreads the 16 pixels of a fragment over the banks
    static constexpr size_t LDS = (size_t)ROWS * COLS * PSTR * sizeof(float) + 16;  // + one dummy float4
};

template <int CIN, int NT, int MT, int RPW, int KS, int ST, bool PLANAR>
__global__ void __launch_bounds__(256) conv2d_kernel(Conv2dParams p) {
    const float amax_seen = absmax_seen(p.absmax);  // read now, used by the epilogue
    using G = KGroup2<CIN>;
    using T = Tile2<CIN, NT, MT, RPW, KS, ST, PLANAR>;
    constexpr int TH = T::TH, TW = T::TW, ROWS = T::ROWS, COLS = T::COLS, PSTR = T::PSTR, PADK = KS / 2;
    extern __shared__ __attribute__((aligned(16))) float slab[];  // [ROWS][COLS][PSTR]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int vox = lane & 15, q = lane >> 4;

    // ---- which tile: each XCD (blockIdx & 7) takes a contiguous band of tiles, so that the halo rows two
    // neighbouring tiles share are served by one L2 ----
    const int ntiles = p.B * p.tiles_h * p.tiles_w;
    int t = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= p.tiles_per_xcd || t >= ntiles) return;  // block-uniform
    const int tw = t % p.tiles_w; t /= p.tiles_w;
    const int th = t % p.tiles_h;
    const int b = t / p.tiles_h;
    const int r0 = th * TH, c0 = tw * TW;
    const int in_r0 = r0 * ST - PADK, in_c0 = c0 * ST - PADK;

    // ---- weight fragments of kernel row kh: [kw][k-group][n-tile][R] per lane ----
    auto load_a = [&](int kh, float (&dst)[KS][G::NKG][NT][G::R]) {
#pragma unroll
        for (int kw = 0; kw < KS; ++kw)
#pragma unroll
            for (int g = 0; g < G::NKG; ++g)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float* wp = p.wpk + ((((size_t)(kh * KS + kw) * G::NKG + g) * NT + n) * 64 + lane) * G::R;
                    if constexpr (G::R == 4) {
                        const float4 v = *reinterpret_cast<const float4*>(wp);
                        dst[kw][g][n][0] = v.x; dst[kw][g][n][1] = v.y; dst[kw][g][n][2] = v.z; dst[kw][g][n][3] = v.w;
                    } else if constexpr (G::R == 2) {
                        const float2 v = *reinterpret_cast<const float2*>(wp);
                        dst[kw][g][n][0] = v.x; dst[kw][g][n][1] = v.y;
                    } else {
                        dst[kw][g][n][0] = *wp;
                    }
                }
    };
    float a_cur[KS][G::NKG][NT][G::R];
    load_a(0, a_cur);  // in flight while the slab is staged

    // ---- stage the input rows + halo, zero outside the image (the conv's zero padding) ----
    float4* __restrict__ slab4 = reinterpret_cast<float4*>(slab);
    if constexpr (PLANAR) {
        // (B,3,hi,wi) image: one slab pixel per element, three coalesced plane reads
        constexpr int NEL = ROWS * COLS, NPF = (NEL + 255) / 256;
        const size_t plane = (size_t)p.hi * p.wi;
        const float* __restrict__ xb = p.x + (size_t)b * 3 * plane;
        float pf[NPF][3];
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int e = tid + 256 * i;
            const int row = e / COLS, col = e - row * COLS;
            const int gr = in_r0 + row, gc = in_c0 + col;
            const bool ok = e < NEL && gr >= 0 && gr < p.hi && gc >= 0 && gc < p.wi;
            const size_t o = ok ? (size_t)gr * p.wi + gc : 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = xb[c * plane + o];
                pf[i][c] = ok ? v : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int e = tid + 256 * i;
            if (e < NEL) slab4[e * (PSTR / 4)] = make_float4(pf[i][0], pf[i][1], pf[i][2], 0.f);
        }
    } else {
        constexpr int C4 = CIN / 4, NEL = ROWS * COLS * C4, NPF = (NEL + 255) / 256;
        const float4* __restrict__ xb4 = reinterpret_cast<const float4*>(p.x) + (size_t)b * p.hi * p.wi * C4;
        float4 pf[NPF];
        bool ok[NPF];
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            // a slab row is one contiguous span of the channel-last input: element `rem` of row `row` sits `rem` float4s
            // behind the row's first pixel — 32-bit index arithmetic, one 64-bit add per load
            const int e = tid + 256 * i;
            const int row = e / (COLS * C4), rem = e - row * (COLS * C4);
            const int gr = in_r0 + row, gc = in_c0 + rem / C4;
            ok[i] = e < NEL && gr >= 0 && gr < p.hi && gc >= 0 && gc < p.wi;
            pf[i] = xb4[ok[i] ? (gr * p.wi + in_c0) * C4 + rem : 0];
        }
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int e = tid + 256 * i;
            const int pix = e / C4, c4 = e - pix * C4;
            if (e < NEL) slab4[pix * (PSTR / 4) + c4] = ok[i] ? pf[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();

    f32x4 acc[RPW][MT][NT];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[r][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int kh = 0; kh < KS; ++kh) {
        float a_nxt[KS][G::NKG][NT][G::R];
        if (kh + 1 < KS) load_a(kh + 1, a_nxt);  // lands during this kernel row's MFMAs
#pragma unroll
        for (int kw = 0; kw < KS; ++kw)
#pragma unroll
            for (int g = 0; g < G::NKG; ++g) {
                float bf[RPW][MT][G::R];
#pragma unroll
                for (int r = 0; r < RPW; ++r)
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const float* bp = slab + (((wave * RPW + r) * ST + kh) * COLS + (m * 16 + vox) * ST + kw) * PSTR +
                                          g * G::KG + G::R * q;
                        if constexpr (G::R == 4) {
                            const float4 v = *reinterpret_cast<const float4*>(bp);
                            bf[r][m][0] = v.x; bf[r][m][1] = v.y; bf[r][m][2] = v.z; bf[r][m][3] = v.w;
                        } else if constexpr (G::R == 2) {
                            const float2 v = *reinterpret_cast<const float2*>(bp);
                            bf[r][m][0] = v.x; bf[r][m][1] = v.y;
                        } else {
                            bf[r][m][0] = *bp;
                        }
                    }
#pragma unroll
                for (int j = 0; j < G::R; ++j)
#pragma unroll
                    for (int r = 0; r < RPW; ++r)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int n = 0; n < NT; ++n)
                                acc[r][m][n] =
                                    __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[kw][g][n][j], bf[r][m][j], acc[r][m][n], 0, 0, 0);
            }
        if (kh + 1 < KS) {
#pragma unroll
            for (int kw = 0; kw < KS; ++kw)
#pragma unroll
                for (int g = 0; g < G::NKG; ++g)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int j = 0; j < G::R; ++j) a_cur[kw][g][n][j] = a_nxt[kw][g][n][j];
        }
    }

    // ---- epilogue: y = act(acc*scale + shift); lane holds couts 16n + 4q .. +3 of pixel column `vox` ----
    float esc[NT][4], esh[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ch = n * 16 + q * 4 + k;
            esc[n][k] = ch < p.Cout ? p.scale[ch] : 0.f;
            esh[n][k] = ch < p.Cout ? p.shift[ch] : 0.f;
        }
    float* __restrict__ yb = p.y + p.oorg + (size_t)b * p.osb;
    float amax = 0.f;  // max |y| of this lane's outputs (p.absmax)
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int orow = r0 + wave * RPW + r;
        if (orow >= p.ho) continue;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int ocol = c0 + m * 16 + vox;
            if (ocol >= p.wo) continue;
            float* __restrict__ yp = yb + (orow * p.osr + ocol * p.osc);  // 32-bit arithmetic, one 64-bit add
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int cb = n * 16 + q * 4;  // first of this lane's 4 output channels
                if (cb >= p.Cout) continue;
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] = fmaf(acc[r][m][n][k], esc[n][k], esh[n][k]);
                    if (p.relu) v[k] = fmaxf(v[k], 0.f);
                    if (p.absmax && cb + k < p.Cout) amax = fmaxf(amax, finite_abs_or_zero(v[k]));
                }
                if (p.osch == 1) {
                    *reinterpret_cast<float4*>(yp + cb) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) yp[(cb + k) * p.osch] = v[k];
                }
            }
        }
    }
    if (p.absmax) {  // what a split-operand layer behind this one scales its activations by; a wave rarely has to raise the slot
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        if ((threadIdx.x & 63) == 0) raise_absmax_seen(p.absmax, amax, amax_seen);
    }
}

template <int CIN, int NT, int MT, int RPW, int KS, int ST, bool PLANAR>
static int launch_conv2d(const Conv2dParams& p0, hipStream_t st) {
    using T = Tile2<CIN, NT, MT, RPW, KS, ST, PLANAR>;
    static_assert(T::LDS <= 160 * 1024, "slab exceeds LDS");
    Conv2dParams p = p0;
    p.tiles_h = (p.ho + T::TH - 1) / T::TH;
    p.tiles_w = (p.wo + T::TW - 1) / T::TW;
    const long long ntiles = (long long)p.B * p.tiles_h * p.tiles_w;
    if (ntiles > 0x3fffffffLL) {
        set_error("conv2d: %lld workgroups exceed the grid limit", ntiles);
        return MVD_ERR_INVALID_ARG;
    }
    p.tiles_per_xcd = (int)((ntiles + 7) / 8);
    auto kern = conv2d_kernel<CIN, NT, MT, RPW, KS, ST, PLANAR>;
    if (T::LDS > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS) != hipSuccess)
        return launch_status("conv2d: LDS attribute");
    hipLaunchKernelGGL(kern, dim3((unsigned)p.tiles_per_xcd * 8), dim3(256), T::LDS, st, p);
    return launch_status("conv2d");
}

// tile shapes per (padded Cin, kernel): slab <= ~60 KB so that two to three workgroups share a CU
template <int CIN, int NT, bool PLANAR>
static int dispatch_kernel_size(const Conv2dParams& p, int ks, int stride, hipStream_t st) {
    if (ks == 3 && stride == 1) {
        constexpr int MT = CIN >= 32 ? 2 : 4;
        return launch_conv2d<CIN, NT, MT, 2, 3, 1, PLANAR>(p, st);
    }
    if (ks == 5 && stride == 2) {
        constexpr int RPW = CIN >= 16 ? 1 : 2;
        return launch_conv2d<CIN, NT, 2, RPW, 5, 2, PLANAR>(p, st);
    }
    set_error("conv2d: kernel %d stride %d unsupported (3/1 or 5/2)", ks, stride);
    return MVD_ERR_INVALID_ARG;
}

template <int CIN, bool PLANAR>
static int dispatch_cout2d(const Conv2dParams& p, int ks, int stride, hipStream_t st) {
    if (p.Cout <= 16) return dispatch_kernel_size<CIN, 1, PLANAR>(p, ks, stride, st);
    return dispatch_kernel_size<CIN, 2, PLANAR>(p, ks, stride, st);
}

static bool cin2d_ok(int c) { return c == 3 || c == 8 || c == 16 || c == 32; }
static bool cout2d_ok(int c) { return c == 8 || c == 16 || c == 32; }
static int padded_cin(int c) { return c == 3 ? 4 : c; }

}  // namespace mvd

extern "C" {

size_t mvd_conv2d_packed_weight_floats(int Cin, int Cout, int ksize) {
    if (!mvd::cin2d_ok(Cin) || !mvd::cout2d_ok(Cout) || (ksize != 3 && ksize != 5)) return 0;
    return (size_t)ksize * ksize * mvd::padded_cin(Cin) * 16 * ((Cout + 15) / 16);
}

int mvd_pack_conv2d_weights_f32(const float* w, int Cin, int Cout, int ksize, float* packed, mvd_stream_t stream) {
    MVD_REQUIRE(w && packed, "pack_conv2d_weights: NULL argument");
    const size_t n = mvd_conv2d_packed_weight_floats(Cin, Cout, ksize);
    MVD_REQUIRE(n > 0, "pack_conv2d_weights: Cin=%d/Cout=%d/k=%d unsupported", Cin, Cout, ksize);
    hipLaunchKernelGGL(mvd::pack_weights2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       packed, Cin, mvd::padded_cin(Cin), Cout, ksize);
    return mvd::launch_status("pack_conv2d_weights");
}

static int conv2d_entry(const float* x, int in_layout, const float* packed_w, const float* scale, const float* shift, float* y, float* y_absmax,
                        int out_layout, int B, int hi, int wi, int Cin, int Cout, int ksize, int stride, int relu, mvd_stream_t stream);

int mvd_conv2d_bn_relu_f32(const float* x, int in_layout, const float* packed_w, const float* scale, const float* shift,
                           float* y, int out_layout, int B, int hi, int wi, int Cin, int Cout, int ksize, int stride,
                           int relu, mvd_stream_t stream) {
    return conv2d_entry(x, in_layout, packed_w, scale, shift, y, nullptr, out_layout, B, hi, wi, Cin, Cout, ksize, stride, relu, stream);
}

int mvd_conv2d_bn_relu_absmax_f32(const float* x, int in_layout, const float* packed_w, const float* scale, const float* shift,
                                  float* y, float* y_absmax, int out_layout, int B, int hi, int wi, int Cin, int Cout, int ksize,
                                  int stride, int relu, mvd_stream_t stream) {
    MVD_REQUIRE(y_absmax, "conv2d_absmax: NULL argument");
    return conv2d_entry(x, in_layout, packed_w, scale, shift, y, y_absmax, out_layout, B, hi, wi, Cin, Cout, ksize, stride, relu, stream);
}

static int conv2d_entry(const float* x, int in_layout, const float* packed_w, const float* scale, const float* shift, float* y, float* y_absmax,
                        int out_layout, int B, int hi, int wi, int Cin, int Cout, int ksize, int stride, int relu, mvd_stream_t stream) {
    MVD_REQUIRE(x && packed_w && scale && shift && y, "conv2d: NULL argument");
    MVD_REQUIRE(B > 0 && hi > 0 && wi > 0, "conv2d: non-positive dimension");
    MVD_REQUIRE((long long)(hi + 3) * (wi + 3) * 32 < 0x7fffffffLL, "conv2d: one %dx%d image exceeds the 32-bit index range", hi, wi);
    MVD_REQUIRE(mvd::cin2d_ok(Cin) && mvd::cout2d_ok(Cout), "conv2d: Cin=%d/Cout=%d unsupported", Cin, Cout);
    MVD_REQUIRE((Cin == 3) == (in_layout == MVD_LAYOUT_NCHW), "conv2d: a 3-channel input must be NCHW and a wider one NHWC");
    MVD_REQUIRE(in_layout == MVD_LAYOUT_NCHW || in_layout == MVD_LAYOUT_NHWC, "conv2d: in_layout=%d unknown", in_layout);
    mvd::Conv2dParams p{};
    p.x = x; p.wpk = packed_w; p.scale = scale; p.shift = shift; p.y = y; p.absmax = y_absmax;
    p.B = B; p.hi = hi; p.wi = wi; p.Cout = Cout; p.relu = relu;
    p.ho = (hi - 1) / stride + 1;  // padding k/2: floor((h + 2*(k/2) - k) / s) + 1
    p.wo = (wi - 1) / stride + 1;
    const long long C = Cout;
    switch (out_layout) {
        case MVD_LAYOUT_NHWC: p.osb = (long long)p.ho * p.wo * C; p.osr = (int)(p.wo * C); p.osc = (int)C; p.osch = 1; p.oorg = 0; break;
        case MVD_LAYOUT_NHWC_BORDER:  // (B, ho+3, wo+3, C) with the image at (1,1): K3's staging layout
            p.osb = (long long)(p.ho + 3) * (p.wo + 3) * C; p.osr = (int)((p.wo + 3) * C); p.osc = (int)C; p.osch = 1; p.oorg = p.osr + C; break;
        case MVD_LAYOUT_NCHW: p.osb = C * p.ho * p.wo; p.osr = p.wo; p.osc = 1; p.osch = p.ho * p.wo; p.oorg = 0; break;
        default: mvd::set_error("conv2d: out_layout=%d unknown", out_layout); return MVD_ERR_INVALID_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (Cin) {
        case 3: return mvd::dispatch_cout2d<4, true>(p, ksize, stride, st);
        case 8: return mvd::dispatch_cout2d<8, false>(p, ksize, stride, st);
        case 16: return mvd::dispatch_cout2d<16, false>(p, ksize, stride, st);
        case 32: return mvd::dispatch_cout2d<32, false>(p, ksize, stride, st);
    }
    return MVD_ERR_INVALID_ARG;
}
}
