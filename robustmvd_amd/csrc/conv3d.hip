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
// (for the transposed conv: of one output parity class, which turns it into 8 small dense convs).
#include "mvd_common.h"

namespace mvd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CONV_TH = 4;
constexpr int CONV_PAD = 4;  // floats of padding per LDS pixel (keeps 16-B alignment, spreads banks)

struct ConvParams {
    const float* x;
    const float* wpk;
    const float* scale;
    const float* shift;
    const float* skip;
    float* y;
    int B, Di, hi, wi;   // input dims
    int Do, ho, wo;      // output dims
    int Cout;
    int relu;
    int tiles_w, tiles_h;  // tiles over the (per-parity) output grid
};

template <int CIN>
struct KGroup {
    static constexpr int KG = CIN >= 16 ? 16 : CIN;  // channels per k-group
    static constexpr int R = KG / 4;                 // consecutive channels per lane = MFMAs per group
    static constexpr int NKG = CIN / KG;
};

// packed weight index: [tap 27][k-group][n-tile][lane 64][R]
template <int CIN>
__host__ __device__ constexpr size_t packed_floats(int nt) {
    return (size_t)27 * KGroup<CIN>::NKG * nt * 64 * KGroup<CIN>::R;
}

template <int CIN>
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cout, int NT,
                                    int transposed) {
    using G = KGroup<CIN>;
    const size_t total = packed_floats<CIN>(NT);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t r = e;
        const int j = r % G::R; r /= G::R;
        const int lane = r % 64; r /= 64;
        const int nt = r % NT; r /= NT;
        const int g = r % G::NKG; r /= G::NKG;
        const int tap = (int)r;
        const int cout = nt * 16 + (lane & 15);
        const int cin = g * G::KG + G::R * (lane >> 4) + j;
        float val = 0.f;
        if (cout < Cout)
            val = transposed ? w[((size_t)cin * Cout + cout) * 27 + tap]   // ConvTranspose3d: (Cin,Cout,3,3,3)
                             : w[((size_t)cout * CIN + cin) * 27 + tap];   // Conv3d: (Cout,Cin,3,3,3)
        packed[e] = val;
    }
}

// MODE: MVD_CONV3D_STRIDE1 / MVD_CONV3D_STRIDE2 / MVD_DECONV3D_STRIDE2
template <int CIN, int NT, int MT, int MODE>
__global__ void __launch_bounds__(256) conv3d_kernel(ConvParams p) {
    using G = KGroup<CIN>;
    constexpr int TW = 16 * MT;
    constexpr int SX = (MODE == MVD_CONV3D_STRIDE2) ? 2 : 1;  // input step per output voxel
    constexpr int ROWS = MODE == MVD_CONV3D_STRIDE1 ? CONV_TH + 2 : MODE == MVD_CONV3D_STRIDE2 ? 2 * CONV_TH + 1 : CONV_TH + 1;
    constexpr int COLS = MODE == MVD_CONV3D_STRIDE1 ? TW + 2 : MODE == MVD_CONV3D_STRIDE2 ? 2 * TW + 1 : TW + 1;
    constexpr int PSTR = CIN + CONV_PAD;  // floats per LDS pixel
    extern __shared__ __attribute__((aligned(16))) float slab[];  // [ROWS][COLS][PSTR]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int vox = lane & 15, q = lane >> 4;

    // ---- which tile ---------------------------------------------------------------------------
    int bx = blockIdx.x;
    const int tw = bx % p.tiles_w; bx /= p.tiles_w;
    const int th = bx % p.tiles_h; bx /= p.tiles_h;
    int par = 0;  // output parity class (deconv only): bit2 = d, bit1 = h, bit0 = w
    if constexpr (MODE == MVD_DECONV3D_STRIDE2) { par = bx & 7; bx >>= 3; }
    const int pd = (par >> 2) & 1, ph = (par >> 1) & 1, pw = par & 1;
    // grid z-extent: output depth planes (conv) or input depth planes (deconv, one per parity class)
    const int nzd = MODE == MVD_DECONV3D_STRIDE2 ? p.Di : p.Do;
    const int zd = bx % nzd;
    const int b = bx / nzd;
    const int r0 = th * CONV_TH;  // first tile row (output rows for conv, input rows a0 for deconv)
    const int c0 = tw * TW;

    // input-plane list: conv: kd = 0..2 -> plane SZ*zd + kd - 1; deconv: pd=0: (k=1, plane zd); pd=1: (k=0, zd+1), (k=2, zd)
    constexpr int SZ = (MODE == MVD_CONV3D_STRIDE2) ? 2 : 1;
    const int nplanes = MODE == MVD_DECONV3D_STRIDE2 ? (pd ? 2 : 1) : 3;
    // first input row / col held by the slab
    const int in_r0 = MODE == MVD_CONV3D_STRIDE1 ? r0 - 1 : MODE == MVD_CONV3D_STRIDE2 ? 2 * r0 - 1 : r0;
    const int in_c0 = MODE == MVD_CONV3D_STRIDE1 ? c0 - 1 : MODE == MVD_CONV3D_STRIDE2 ? 2 * c0 - 1 : c0;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float4* __restrict__ wpk4 = reinterpret_cast<const float4*>(p.wpk);
    (void)wpk4;

    for (int ip = 0; ip < nplanes; ++ip) {
        int kd, plane;
        if constexpr (MODE == MVD_DECONV3D_STRIDE2) {
            kd = pd ? (ip == 0 ? 0 : 2) : 1;
            plane = pd ? (ip == 0 ? zd + 1 : zd) : zd;
        } else {
            kd = ip;
            plane = SZ * zd + ip - 1;
        }
        const bool plane_ok = plane >= 0 && plane < p.Di;  // block-uniform
        if (!plane_ok) continue;                            // contributes zeros

        __syncthreads();  // previous plane's reads are done
        // ---- stage the input rows of this plane (zero-filled halo) ----------------------------
        {
            constexpr int C4 = CIN / 4;
            constexpr int NV = ROWS * COLS * C4;
            const float* __restrict__ xp = p.x + ((size_t)b * p.Di + plane) * p.hi * p.wi * CIN;
            for (int e = tid; e < NV; e += 256) {
                const int c4 = e % C4;
                const int col = (e / C4) % COLS;
                const int row = e / (C4 * COLS);
                const int gr = in_r0 + row, gc = in_c0 + col;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gr >= 0 && gr < p.hi && gc >= 0 && gc < p.wi)
                    v = *reinterpret_cast<const float4*>(xp + ((size_t)gr * p.wi + gc) * CIN + c4 * 4);
                *reinterpret_cast<float4*>(slab + (row * COLS + col) * PSTR + c4 * 4) = v;
            }
        }
        __syncthreads();

        // ---- taps of this plane ----------------------------------------------------------------
        const int nkh = MODE == MVD_DECONV3D_STRIDE2 ? (ph ? 2 : 1) : 3;
        const int nkw = MODE == MVD_DECONV3D_STRIDE2 ? (pw ? 2 : 1) : 3;
        for (int ih = 0; ih < nkh; ++ih) {
            int kh, srow;  // kernel index, slab row read by this wave
            if constexpr (MODE == MVD_DECONV3D_STRIDE2) {
                kh = ph ? (ih == 0 ? 0 : 2) : 1;
                srow = wave + (ph ? (ih == 0 ? 1 : 0) : 0);
            } else {
                kh = ih;
                srow = SX * wave + ih;
            }
            for (int iw = 0; iw < nkw; ++iw) {
                int kw, scol;  // kernel index, slab column of output voxel 0 of the tile
                if constexpr (MODE == MVD_DECONV3D_STRIDE2) {
                    kw = pw ? (iw == 0 ? 0 : 2) : 1;
                    scol = pw ? (iw == 0 ? 1 : 0) : 0;
                } else {
                    kw = iw;
                    scol = iw;
                }
                const int tap = (kd * 3 + kh) * 3 + kw;
                const float* __restrict__ srow_p = slab + (srow * COLS + scol) * PSTR + G::R * q;
#pragma unroll
                for (int g = 0; g < G::NKG; ++g) {
                    // A fragments (weights): R floats per lane, contiguous per (tap, g, nt)
                    float af[NT][G::R];
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const float* wp = p.wpk + ((((size_t)tap * G::NKG + g) * NT + n) * 64 + lane) * G::R;
                        if constexpr (G::R == 4) {
                            const float4 t = *reinterpret_cast<const float4*>(wp);
                            af[n][0] = t.x; af[n][1] = t.y; af[n][2] = t.z; af[n][3] = t.w;
                        } else {
                            const float2 t = *reinterpret_cast<const float2*>(wp);
                            af[n][0] = t.x; af[n][1] = t.y;
                        }
                    }
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const float* bp = srow_p + ((m * 16 + vox) * SX) * PSTR + g * G::KG;
                        float bf[G::R];
                        if constexpr (G::R == 4) {
                            const float4 t = *reinterpret_cast<const float4*>(bp);
                            bf[0] = t.x; bf[1] = t.y; bf[2] = t.z; bf[3] = t.w;
                        } else {
                            const float2 t = *reinterpret_cast<const float2*>(bp);
                            bf[0] = t.x; bf[1] = t.y;
                        }
#pragma unroll
                        for (int n = 0; n < NT; ++n)
#pragma unroll
                            for (int j = 0; j < G::R; ++j)
                                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[n][j], bf[j], acc[m][n], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue: y = act(acc*scale + shift) (+ skip); lane holds couts 16n + 4q .. +3 of voxel `vox` ----
    int orow, ocol_step, ocol0, oz;
    if constexpr (MODE == MVD_DECONV3D_STRIDE2) {
        orow = 2 * (r0 + wave) + ph;
        ocol0 = 2 * c0 + pw;
        ocol_step = 2;
        oz = 2 * zd + pd;
    } else {
        orow = r0 + wave;
        ocol0 = c0;
        ocol_step = 1;
        oz = zd;
    }
    if (orow >= p.ho) return;
    const size_t row_base = (((size_t)b * p.Do + oz) * p.ho + orow) * p.wo;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int ocol = ocol0 + (m * 16 + vox) * ocol_step;
        if (ocol >= p.wo) continue;
        const size_t o = (row_base + ocol) * p.Cout;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int cb = n * 16 + q * 4;
            if (cb >= p.Cout) continue;
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = cb + k;
                float val = 0.f;
                if (c < p.Cout) {
                    val = fmaf(acc[m][n][k], p.scale[c], p.shift[c]);
                    if (p.relu) val = fmaxf(val, 0.f);
                    if (p.skip) val += p.skip[o + c];
                }
                r[k] = val;
            }
            if (cb + 3 < p.Cout) {
                *reinterpret_cast<float4*>(p.y + o + cb) = make_float4(r[0], r[1], r[2], r[3]);
            } else {
                for (int k = 0; k < 4; ++k)
                    if (cb + k < p.Cout) p.y[o + cb + k] = r[k];
            }
        }
    }
}

template <int CIN, int NT, int MT, int MODE>
static int launch_conv(const ConvParams& p0, hipStream_t st) {
    ConvParams p = p0;
    constexpr int TW = 16 * MT;
    constexpr int ROWS = MODE == MVD_CONV3D_STRIDE1 ? CONV_TH + 2 : MODE == MVD_CONV3D_STRIDE2 ? 2 * CONV_TH + 1 : CONV_TH + 1;
    constexpr int COLS = MODE == MVD_CONV3D_STRIDE1 ? TW + 2 : MODE == MVD_CONV3D_STRIDE2 ? 2 * TW + 1 : TW + 1;
    constexpr size_t lds = (size_t)ROWS * COLS * (CIN + CONV_PAD) * sizeof(float);
    static_assert(lds <= 160 * 1024, "slab exceeds LDS");
    // tiles over the output grid (conv) or over the input grid = one parity class of the output (deconv)
    const int gh = MODE == MVD_DECONV3D_STRIDE2 ? p.hi : p.ho;
    const int gw = MODE == MVD_DECONV3D_STRIDE2 ? p.wi : p.wo;
    p.tiles_h = (gh + CONV_TH - 1) / CONV_TH;
    p.tiles_w = (gw + TW - 1) / TW;
    const long long nz = MODE == MVD_DECONV3D_STRIDE2 ? (long long)p.Di * 8 : p.Do;
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

template <int CIN, int MODE>
static int dispatch_cout(const ConvParams& p, hipStream_t st) {
    // MT (16-voxel tiles per wave) chosen so that the slab fits LDS and wide rows get long tiles
    constexpr int MT = MODE == MVD_CONV3D_STRIDE2 ? (CIN >= 32 ? 2 : 4) : 4;
    const int nt = (p.Cout + 15) / 16;
    switch (nt) {
        case 1: return launch_conv<CIN, 1, MT, MODE>(p, st);
        case 2: return launch_conv<CIN, 2, MT, MODE>(p, st);
        case 4: return launch_conv<CIN, 4, MT, MODE>(p, st);
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
    return (size_t)27 * Cin * 16 * ((Cout + 15) / 16);
}

int mvd_pack_conv3d_weights_f32(const float* w, int Cin, int Cout, int mode, float* packed, mvd_stream_t stream) {
    MVD_REQUIRE(w && packed, "pack_conv3d_weights: NULL argument");
    MVD_REQUIRE(mvd::cin_ok(Cin) && mvd::cout_ok(Cout), "pack_conv3d_weights: Cin=%d/Cout=%d unsupported", Cin, Cout);
    MVD_REQUIRE(mode >= 0 && mode <= 2, "pack_conv3d_weights: mode=%d unknown", mode);
    const int NT = (Cout + 15) / 16;
    const int tr = mode == MVD_DECONV3D_STRIDE2;
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = (unsigned)((mvd_conv3d_packed_weight_floats(Cin, Cout) + 255) / 256);
    switch (Cin) {
        case 8: hipLaunchKernelGGL(mvd::pack_weights_kernel<8>, dim3(nb), dim3(256), 0, st, w, packed, Cout, NT, tr); break;
        case 16: hipLaunchKernelGGL(mvd::pack_weights_kernel<16>, dim3(nb), dim3(256), 0, st, w, packed, Cout, NT, tr); break;
        case 32: hipLaunchKernelGGL(mvd::pack_weights_kernel<32>, dim3(nb), dim3(256), 0, st, w, packed, Cout, NT, tr); break;
        case 64: hipLaunchKernelGGL(mvd::pack_weights_kernel<64>, dim3(nb), dim3(256), 0, st, w, packed, Cout, NT, tr); break;
    }
    return mvd::launch_status("pack_conv3d_weights");
}

int mvd_conv3d_bn_relu_f32(const float* x, const float* packed_w, const float* scale, const float* shift,
                           const float* skip, float* y, int B, int Di, int hi, int wi, int Cin, int Cout, int mode,
                           int relu, mvd_stream_t stream) {
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
    switch (Cin) {
        case 8: return mvd::dispatch_mode<8>(p, mode, st);
        case 16: return mvd::dispatch_mode<16>(p, mode, st);
        case 32: return mvd::dispatch_mode<32>(p, mode, st);
        case 64: return mvd::dispatch_mode<64>(p, mode, st);
    }
    return MVD_ERR_INVALID_ARG;
}
}
