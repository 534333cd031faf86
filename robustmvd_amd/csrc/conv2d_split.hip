// Path A's 2-D CNN (the DispNet encoder / context encoder / fusion score convs / cost-volume encoder / decoder of robust_mvd:
// rmvd/models/blocks/dispnet_encoder.py:6-27, dispnet_context_encoder.py, learned_fusion.py:8-20,
// dispnet_costvolume_encoder.py:7-50, dispnet_decoder.py:36-138) as ONE implicit-GEMM kernel family on fp16 MFMA with
// split fp32 operands: fp32-grade results at several times the fp32 matrix rate, no layout transposes, no im2col, no `cat`.
//
// Arithmetic (the 2-D form of conv3d_split.hip): every fp32 activation a and weight w becomes two fp16 terms,
//     a 2^-e = a_hi + a_lo,  a_hi = fp16(a 2^-e),  a_lo = fp16(a 2^-e - a_hi)            (same for w with 2^-k_c per cout)
// and a w is evaluated as a_hi w_hi + a_hi w_lo + a_lo w_hi on v_mfma_f32_16x16x32_f16 with ONE fp32 accumulator (the lo terms
// are not rescaled here: with the block scaling below they stay in fp16's range, denormals included).  e puts max |x| of the
// whole input tensor into [2^14, 2^15), k_c puts max |w_c| into [2^10, 2^11); the epilogue multiplies by 2^(e + k_c).  A value v
// is represented to |error| <= max(2^-22 |v|, 2^-39 max|x|).  max |x| comes from the producer's epilogue (every kernel here can
// leave max |y| behind: one atomicMax per workgroup) or from mvd_absmax_f32.
//
// Data layout: activations NHWC fp32 with a free pixel stride (so that a layer can read or write a channel slice of a wider
// buffer: the decoder's concat inputs are never copied together), channel count a multiple of 8 (pad channels hold zeros and
// meet zero weights).  Weights pre-split and pre-packed in MFMA fragment order.
//
// GEMM view: D[cout][pixel] += W[cout][k] X[k][pixel], k = (tap, channel).  A workgroup (4 waves) owns a 16 x 16 tile of output
// pixels and BN output channels.  Per chunk of CC = 8 U8 input channels the input patch the tile needs is staged ONCE in LDS
// (converted to the two fp16 terms on the way; zero outside the image = the padding), stride-2 layers de-interleave the
// columns by parity so that the 16 pixels of a fragment read conflict-free.  A K step = 4 units of 8 channels (unit = (tap,
// 8-channel group)): per step a wave reads 2 activation fragments per pixel row from LDS and its weight fragments straight from
// L2 (each wave owns its own output channels, so no weight byte is loaded twice per workgroup), 3 MFMAs per 16 x 16 block.
// Layers with few pixels and many weights (conv5 .. deconv_2) split K over workgroups; partial sums go to a workspace and a
// second kernel adds them in a fixed order (run-to-run identical) and applies the epilogue.
//
// Transposed convolutions (4 x 4, stride 2, padding 1) run as four 2 x 2 stride-1 layers, one per output parity class.
#include "mvd_common.h"
#include <algorithm>
#include <stdlib.h>

// knock-out builds for tools/ko_conv2d.sh (timing only, WRONG results): 1 stage only a workgroup's first chunk, 2 no MFMAs,
// 4 weights loaded once, 8 activation fragments read once per step, 16 no stores
#ifndef C2_KO
#define C2_KO 0
#endif

namespace mvd {

typedef _Float16 c2h8 __attribute__((ext_vector_type(8)));
typedef float c2f4 __attribute__((ext_vector_type(4)));
typedef unsigned int c2u4 __attribute__((ext_vector_type(4)));
constexpr int C2_TH = 16, C2_TW = 16;  // output pixels per workgroup tile

template <int KH, int KW, int S, int U8, bool IMG = false>
struct C2Geom {
    static constexpr int CC = 8 * U8;                          // input channels per staged chunk
    static constexpr int PB = 16;                              // LDS bytes per pixel, term and 8-channel group
    static constexpr int PH = (C2_TH - 1) * S + KH, PW = (C2_TW - 1) * S + KW;  // staged patch
    static constexpr int PWS = (PW + S - 1) / S;               // columns per parity plane
    static constexpr int ROWB = S * S * PWS * PB;              // LDS bytes between the patch rows of consecutive OUTPUT rows
    // The 8-channel groups of a chunk are PLANES C8S bytes apart, C8S a multiple of the 256-byte bank row: `ds_read_b128` serves the
    // lanes in four NON-contiguous groups of 16 ({0-3, 12-15, 20-27}, ...), i.e. 8 pixels of one K group of the fragment and the
    // other 8 pixels of the next; with the K groups (= channel groups, when a chunk has 4) a whole number of bank rows apart those
    // 16 lanes read 16 different 16-byte slots.  (Pixel-interleaved groups at an odd pitch of 80 bytes kept ONE K group conflict-free
    // but not the hardware's groups: half of the LDS cycles were bank conflicts, profiles/r03_conv2d_fusion3x3_pmc.txt.)
    static constexpr int C8S = (PH * S * PWS * PB + 255) / 256 * 256;
    static constexpr int PLANE = U8 * C8S;                     // one term of the patch
    static constexpr int UNITS = KH * KW * U8;                 // (tap, 8-channel group) units per chunk
    static constexpr int STEPS = (UNITS + 3) / 4;              // MFMA K steps (32 = 4 units) per chunk
    static constexpr int ITEMS = PH * PW * U8;                 // staging items (pixel, 8-channel group)
    static constexpr int NIT = (ITEMS + 255) / 256;
    static_assert(2 * PLANE <= 80 * 1024, "patch exceeds half the LDS (two workgroups per CU)");
};
// The first layer on a planar 3-channel image (7 x 7, stride 2): a pixel is 4 halves (r, g, b, 0) = 8 bytes, rows are stored as
// they are (no parity planes), so the 16 bytes a lane reads are the TWO x-adjacent pixels of taps kx = 2g, 2g + 1 and a K step is
// one kernel row: 7 x 4 = 28 real values of 32 (kx = 7 meets zero weights) in 7 steps, against 13 steps of 8-channel pixels.
template <int KH, int KW, int S, int U8>
struct C2Geom<KH, KW, S, U8, true> {
    static_assert(KH == 7 && KW == 7 && S == 2 && U8 == 1, "image layer: 7 x 7, stride 2");
    static constexpr int CC = 8, PB = 8;
    static constexpr int PH = (C2_TH - 1) * S + KH, PW = (C2_TW - 1) * S + KW + 1;  // + the column tap kx = 7 reads
    static constexpr int ROWB = S * PW * PB;
    static constexpr int PLANE = PH * PW * PB;
    static constexpr int STEPS = KH;
    static constexpr int ITEMS = PH * PW;
    static constexpr int NIT = (ITEMS + 255) / 256;
    static_assert(PW % 2 == 0, "16-byte aligned pixel pairs");
};

struct C2Params {
    const float* x;       // input, NHWC, first channel of the slice; pixel stride xs floats.  NCHW3: (B, 3, Hi, Wi) planes
    const float* xamax;   // max |x| (device, one float)
    const char* wpk;      // packed weight fragments (transposed conv: 4 parity classes, cls_bytes apart)
    const float* eun;     // per output channel 2^k_c (padded to a multiple of 16)
    const float* bias;    // Cout or NULL (3-D layers: the folded batch-norm shift)
    const float* scale;   // NULL, or the folded batch-norm scale per channel: y = act(conv * scale + bias)
    const float* skip;    // NULL, or a tensor laid out like y that is added after the activation
    float* y;             // output, NHWC, first channel of the slice; pixel stride ys floats
    float* yamax;         // optional: max |y| over the finite outputs (atomicMax; zeroed by the caller)
    float* part;          // split-K: partial sums [ksplit][B Ho Wo][ncp]; NULL = direct epilogue
    int B, Hi, Wi, xs, nchunks;   // B = output images of the launch: batch x output planes for a 3-D layer
    // 3-D layers (a 2-D layer has Di = Do = KD = SD = 1, pad_d = 0): image b' = (b, od) reads input planes od * SD - pad_d + kd;
    // chunk = kd * nchunks_c + c.  sub3d: a transposed 3 x 3 x 3 stride-2 layer run as a 2 x 2 x 2 convolution whose output channels
    // are (parity class (ad, ay, ax), channel): class goes to output voxel (2 od + ad, 2 oy + ay, 2 ox + ax)
    int Di, Do, Dy, KD, SD, pad_d, nchunks_c, sub3d, Cr;
    int Ho, Wo;           // output grid of this launch (transposed conv: one parity class = the input grid)
    int Hy, Wy, ys;       // the output tensor's full grid; floats between pixels
    long long ys_row, ys_img, ycs;  // floats between output rows, images and channels (ycs = 1: channel-last)
    int oy_mul, ox_mul;   // output pixel (oy, ox) of the grid lands at (oy * oy_mul + a, ox * ox_mul + b)
    int pad_y, pad_x;     // input row of tap ky for output row oy: oy * S - pad_y + ky
    int Cout, ncp;        // output channels; ncp = Cout padded to the launch's BN
    int tiles_x, tiles_y, nblocks, ksplit, chunks_per_split;
    int ncls;             // 1, or 4 = the output parity classes (a, b) of a transposed conv: pad - (a, b), output offset (a, b)
    long long cls_bytes;
    int act;              // 0 none, 1 LeakyReLU(slope), 2 ReLU
    float slope;
};

// per output channel 2^k_c, k_c = exponent(max |w_c|) - 10 (0 for an all-zero channel); Conv2d (Cout, Cin, KH, KW) or
// ConvTranspose2d (Cin, Cout, KH, KW) layout
__global__ void c2_wscale_kernel(const float* __restrict__ w, float* __restrict__ eun, int Cin, int Cout, int taps, int transposed, int cpad) {
    __shared__ float red[256];
    const int c = blockIdx.x;
    float m = 0.f;
    if (c < Cout)
        for (int e = threadIdx.x; e < Cin * taps; e += 256) {
            const int ci = e / taps, t = e % taps;
            const float v = fabsf(transposed ? w[((size_t)ci * Cout + c) * taps + t] : w[((size_t)c * Cin + ci) * taps + t]);
            m = (v <= 3.4e38f && v > m) ? v : m;
        }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0 && c < cpad) {
        const int ex = (int)((__float_as_uint(red[0]) >> 23) & 0xffu) - 127;
        const int k = red[0] > 0.f ? max(-100, min(100, ex - 10)) : 0;
        eun[c] = ldexpf(1.0f, k);
    }
}

// -> [class][cout tile (16)][chunk][step][term hi, lo][lane 64][8 halves]; lane l: cout 16 nt + l % 16, unit 4 step + l / 16,
// unit u = (tap u / U8, 8-channel group u % U8), channel = chunk CC + 8 (u % U8) + j.  Units past the taps, channels past Cin
// and output channels past Cout get zeros.  Transposed (4 x 4, stride 2, padding 1) class (a, b): tap (ty, tx) of the 2 x 2
// layer = kernel element (3 - a - 2 ty, 3 - b - 2 tx).
__global__ void c2_pack_kernel(const float* __restrict__ w, const float* __restrict__ eun, _Float16* __restrict__ packed, int Cin, int Cout,
                               int KH, int KW, int U8, int nchunks, int ntiles, int steps, int transposed, long long total) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63), term = (int)((e >> 9) & 1);
    long long r = e >> 10;
    const int step = (int)(r % steps); r /= steps;
    const int chunk = (int)(r % nchunks); r /= nchunks;
    const int nt = (int)(r % ntiles);
    const int cls = (int)(r / ntiles);
    const int cout = nt * 16 + (lane & 15), unit = 4 * step + (lane >> 4);
    int tap = unit / U8, cin = chunk * 8 * U8 + 8 * (unit % U8) + j;
    if (transposed == 2) {  // image layer: step = kernel row, lane group = the tap pair kx = 2g, 2g + 1, 4 halves per pixel
        const int kx = 2 * (lane >> 4) + j / 4;
        tap = kx < KW ? step * KW + kx : KH * KW;
        cin = j % 4;
    }
    float v = 0.f;
    if (tap < KH * KW && cin < Cin && cout < Cout) {
        const int ty = tap / KW, tx = tap % KW;
        if (transposed == 1) {
            const int a = cls >> 1, b = cls & 1, ky = 3 - a - 2 * ty, kx = 3 - b - 2 * tx;
            v = w[(((size_t)cin * Cout + cout) * 4 + ky) * 4 + kx];
        } else {
            v = w[(((size_t)cout * Cin + cin) * KH + ty) * KW + tx];
        }
        v = v / eun[cout];
    }
    const _Float16 hi = (_Float16)v;
    packed[e] = term == 0 ? hi : (_Float16)(v - (float)hi);
}

// epilogue of 4 consecutive output channels cb .. cb + 3 of output pixel (oy, ox) of image bi (main kernel and split-K reduce)
// the per-channel constants of output channels cb .. cb + 3 (loaded once per channel group, not once per pixel)
struct C2Chan {
    float mul[4], sc[4], bi[4];  // 2^(e + k_c); batch-norm scale (1 when there is none); bias / shift
};
__device__ __forceinline__ C2Chan c2_chan(const C2Params& p, int cb, float xsc_inv) {
    C2Chan k;
    const int c = p.sub3d ? cb % p.Cr : cb;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool in = cb + r < p.Cout;
        k.mul[r] = in ? p.eun[cb + r] * xsc_inv : 0.f;
        k.sc[r] = (in && p.scale) ? p.scale[c + r] : 1.f;
        k.bi[r] = (in && p.bias) ? p.bias[c + r] : 0.f;
    }
    return k;
}
// offset (floats) of output channel cb of output pixel (oy, ox) of image (b, od); consecutive oy are c2_row_step(p) apart
__device__ __forceinline__ size_t c2_out_offset(const C2Params& p, int b, int od, int oy, int ox, int cls, int cb) {
    int c = cb, zo = od, yo = oy * p.oy_mul + (cls >> 1), xo = ox * p.ox_mul + (cls & 1);
    if (p.sub3d) {
        const int k3 = cb / p.Cr;
        c = cb - k3 * p.Cr;
        zo = 2 * od + (k3 >> 2); yo = 2 * oy + ((k3 >> 1) & 1); xo = 2 * ox + (k3 & 1);
    }
    return ((size_t)b * p.Dy + zo) * p.ys_img + (size_t)yo * p.ys_row + (size_t)xo * p.ys + (size_t)c * p.ycs;
}
__device__ __forceinline__ size_t c2_row_step(const C2Params& p) { return (size_t)(p.sub3d ? 2 : p.oy_mul) * p.ys_row; }
// epilogue of 4 consecutive output channels cb .. cb + 3 at output offset o (main kernel and split-K reduce)
__device__ __forceinline__ void c2_finish(const C2Params& p, size_t o, int cb, c2f4 a, const C2Chan& k, float& amax) {
    float r4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = 0.f;
        if (cb + r < p.Cout) {
            const float conv = a[r] * k.mul[r];  // exact: a power of two
            v = p.scale ? fmaf(conv, k.sc[r], k.bi[r]) : conv + k.bi[r];
            if (p.act == 1) v = v > 0.f ? v : v * p.slope;
            else if (p.act == 2) v = fmaxf(v, 0.f);
            if (p.skip) v += p.skip[o + r * p.ycs];
            amax = fmaxf(amax, finite_abs_or_zero(v));
        }
        r4[r] = v;
    }
    if (cb + 3 < p.Cout && p.ycs == 1) {
        *reinterpret_cast<c2f4*>(p.y + o) = c2f4{r4[0], r4[1], r4[2], r4[3]};
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (cb + r < p.Cout) p.y[o + r * p.ycs] = r4[r];
    }
}

template <int KH, int KW, int S, int U8, int WM, int NTW, bool NCHW3>
__global__ void __launch_bounds__(256, 2) conv2d_split_kernel(C2Params p) {
    const float amax_seen = absmax_seen(p.part ? nullptr : p.yamax);  // read now, used by the epilogue
    using G = C2Geom<KH, KW, S, U8, NCHW3>;
    constexpr int WN = 4 / WM, MTW = 16 / WM, STEPS = G::STEPS, PB = G::PB, PLANE = G::PLANE;
    // (Measured and dropped: the next chunk's global reads sent to a per-thread LDS scratch by LDS-DMA under this chunk's MFMAs.
    // vmcnt retires in order, so the weight fragments of the next K step then wait for those slow reads: 5-10 % slower.)
    extern __shared__ __attribute__((aligned(16))) char patch[];  // [term][row][column parity][column / S][PB] | 4 floats
    float* const wmax = reinterpret_cast<float*>(patch + 2 * PLANE);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int g = lane >> 4, px16 = lane & 15;

    int bx = blockIdx.x;
    const int nb = bx % p.nblocks; bx /= p.nblocks;
    const int ks = bx % p.ksplit; bx /= p.ksplit;
    const int cls = bx % p.ncls; bx /= p.ncls;
    const int tx = bx % p.tiles_x; bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int b = bx / p.tiles_y;               // output image: (batch element, output plane) for a 3-D layer
    const int od = b % p.Do, bb = b / p.Do;
    const int ca = cls >> 1, cb2 = cls & 1;  // (0, 0) for a plain convolution
    const int oy0 = ty * C2_TH, ox0 = tx * C2_TW;
    const int iy0 = oy0 * S - (p.pad_y - ca), ix0 = ox0 * S - (p.pad_x - cb2);

    // activation scale 2^-e, e = exponent(max |x|) - 14
    float xsc, xsc_inv;
    {
        const unsigned mb = __builtin_amdgcn_readfirstlane((int)__float_as_uint(*p.xamax));
        const int ex = (int)((mb >> 23) & 0xffu) - 127;
        const int e = max(-125, min(125, ex - 14));
        xsc = __uint_as_float((unsigned)(127 - e) << 23);
        xsc_inv = __uint_as_float((unsigned)(127 + e) << 23);
    }
    const float one = 1.0f;
    auto split2 = [xsc, one](float a0, float a1, unsigned& hi, unsigned& lo) {
        unsigned hp, lp;
        float t0, t1;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(hp) : "v"(a0), "s"(xsc));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(hp) : "v"(a1), "s"(xsc));
        asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(t0) : "v"(a0), "s"(xsc), "v"(hp));
        asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(t1) : "v"(a1), "s"(xsc), "v"(hp));
        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(lp) : "v"(t0), "s"(one));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(lp) : "v"(t1), "s"(one));
        hi = hp;
        lo = lp;
    };

    // staging items of this thread: global offset (floats, without the chunk's channel offset; -1 = outside the image) and LDS byte
    long long goff[G::NIT];
    int loff[G::NIT];
#pragma unroll
    for (int k = 0; k < G::NIT; ++k) {
        const int it = tid + 256 * k;
        const int pix = it / U8, c8 = it % U8;
        const int py = pix / G::PW, pxx = pix % G::PW;
        const int iy = iy0 + py, ix = ix0 + pxx;
        const bool live = it < G::ITEMS;
        const bool inside = live && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        if constexpr (NCHW3) {
            goff[k] = inside ? ((long long)b * 3 * p.Hi + iy) * p.Wi + ix : -1;
            loff[k] = live ? pix * PB : -1;
        } else {  // inside its input image; the image (plane) is the chunk's
            goff[k] = inside ? ((long long)iy * p.Wi + ix) * p.xs + c8 * 8 : -1;
            loff[k] = live ? ((py * S + pxx % S) * G::PWS + pxx / S) * PB + c8 * G::C8S : -1;
        }
    }
    // activation fragment address of this lane per K step (row 0 of the wave's rows)
    int tapoff[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        if constexpr (NCHW3) {  // step = kernel row, lane group g = the tap pair kx = 2g, 2g + 1
            tapoff[s] = (s * G::PW + 2 * px16 + 2 * g) * PB + wm * MTW * G::ROWB;
        } else {
            int unit = 4 * s + g;
            if (unit >= G::UNITS) unit = 0;  // meets zero weights
            const int tap = unit / U8, c8 = unit % U8, ky = tap / KW, kx = tap % KW;
            tapoff[s] = ((ky * S + kx % S) * G::PWS + kx / S + px16) * PB + c8 * G::C8S + wm * MTW * G::ROWB;
        }
    }

    // weights: this wave's cout tiles
    const int nt0 = nb * (WN * NTW) + wn * NTW;
    const size_t nt_stride = (size_t)p.nchunks * STEPS * 2048;
    const char* wlane = p.wpk + (size_t)cls * p.cls_bytes + (size_t)nt0 * nt_stride + lane * 16;

    c2f4 acc[MTW][NTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[m][t] = c2f4{0, 0, 0, 0};

    const int c_begin = ks * p.chunks_per_split, c_end = min(p.nchunks, c_begin + p.chunks_per_split);

    // weight fragments one K step ahead (steps and chunks are consecutive in the packed buffer); the last prefetch re-reads
    c2h8 wh[NTW], wl[NTW];
    const char* wc = wlane + (size_t)c_begin * STEPS * 2048;
    const char* const wlast = wlane + ((size_t)max(c_end, c_begin + 1) * STEPS - 1) * 2048;
    if (c_begin < c_end) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            wh[t] = *reinterpret_cast<const c2h8*>(wc + t * nt_stride);
            wl[t] = *reinterpret_cast<const c2h8*>(wc + t * nt_stride + 1024);
        }
    }

    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        // ---- stage the chunk's patch: global fp32 -> two fp16 terms -> LDS -------------------------------------------------------
        if (!(C2_KO & 1) || chunk == c_begin) {
            c2f4 v0[G::NIT], v1[G::NIT];
            const int kd = chunk / p.nchunks_c, zi = od * p.SD - p.pad_d + kd;  // the input plane of this chunk (0 for a 2-D layer)
            const bool zlive = zi >= 0 && zi < p.Di;
            const float* xim = p.x + ((size_t)bb * p.Di + (zlive ? zi : 0)) * p.Hi * p.Wi * p.xs + (size_t)(chunk - kd * p.nchunks_c) * G::CC;
#pragma unroll
            for (int k = 0; k < G::NIT; ++k) {
                v0[k] = c2f4{0, 0, 0, 0};
                v1[k] = c2f4{0, 0, 0, 0};
                if (goff[k] >= 0 && zlive) {
                    if constexpr (NCHW3) {
                        const size_t pl = (size_t)p.Hi * p.Wi;
                        v0[k][0] = p.x[goff[k]]; v0[k][1] = p.x[goff[k] + pl]; v0[k][2] = p.x[goff[k] + 2 * pl];
                    } else {
                        v0[k] = *reinterpret_cast<const c2f4*>(xim + goff[k]);
                        v1[k] = *reinterpret_cast<const c2f4*>(xim + goff[k] + 4);
                    }
                }
            }
            __syncthreads();  // every wave has finished reading the previous chunk's patch
#pragma unroll
            for (int k = 0; k < G::NIT; ++k) {
                if (loff[k] < 0) continue;
                unsigned h0, h1, h2, h3, l0, l1, l2, l3;
                split2(v0[k][0], v0[k][1], h0, l0);
                split2(v0[k][2], v0[k][3], h1, l1);
                if constexpr (NCHW3) {
                    *reinterpret_cast<unsigned long long*>(patch + loff[k]) = (unsigned long long)h0 | ((unsigned long long)h1 << 32);
                    *reinterpret_cast<unsigned long long*>(patch + loff[k] + PLANE) = (unsigned long long)l0 | ((unsigned long long)l1 << 32);
                    continue;
                }
                split2(v1[k][0], v1[k][1], h2, l2);
                split2(v1[k][2], v1[k][3], h3, l3);
                *reinterpret_cast<c2u4*>(patch + loff[k]) = c2u4{h0, h1, h2, h3};
                *reinterpret_cast<c2u4*>(patch + loff[k] + PLANE) = c2u4{l0, l1, l2, l3};
            }
            __syncthreads();
        }

        // ---- K steps: activation fragments one pixel row ahead, weights one step ahead ---------------------------------------------
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            c2h8 nh[NTW], nl[NTW];
            if (C2_KO & 4) {
#pragma unroll
                for (int t = 0; t < NTW; ++t) { nh[t] = wh[t]; nl[t] = wl[t]; }
            } else {
                const char* wn = wc + 2048;
                wn = wn > wlast ? wlast : wn;
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    nh[t] = *reinterpret_cast<const c2h8*>(wn + t * nt_stride);
                    nl[t] = *reinterpret_cast<const c2h8*>(wn + t * nt_stride + 1024);
                }
                wc += 2048;
            }
            const char* a = patch + tapoff[s];
            // activation fragments TWO pixel rows ahead (one row = 3 NTW MFMAs = 48 .. 96 clocks: less than a conflicted LDS read)
            c2h8 xh = *reinterpret_cast<const c2h8*>(a), xl = *reinterpret_cast<const c2h8*>(a + PLANE);
            c2h8 yh = xh, yl = xl;
            if (MTW > 1) {
                yh = *reinterpret_cast<const c2h8*>(a + G::ROWB);
                yl = *reinterpret_cast<const c2h8*>(a + G::ROWB + PLANE);
            }
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const c2h8 ch = xh, cl = xl;
                xh = yh; xl = yl;
                if (m + 2 < MTW && !(C2_KO & 8)) {
                    yh = *reinterpret_cast<const c2h8*>(a + (m + 2) * G::ROWB);
                    yl = *reinterpret_cast<const c2h8*>(a + (m + 2) * G::ROWB + PLANE);
                }
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    if (C2_KO & 2) {
                        acc[m][t][0] += (float)ch[0] * (float)wh[t][0] + (float)cl[1] * (float)wl[t][1];
                        continue;
                    }
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], ch, acc[m][t], 0, 0, 0);
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t], ch, acc[m][t], 0, 0, 0);
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], cl, acc[m][t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);  // keeps the compiler from hoisting every row's reads to the top (registers)
            }
#pragma unroll
            for (int t = 0; t < NTW; ++t) { wh[t] = nh[t]; wl[t] = nl[t]; }
        }
    }

    // ---- epilogue: lane holds output channels cb .. cb + 3 of pixel (row m, column px16) -------------------------------------------
    const int ox = ox0 + px16;
    float amax = 0.f;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int cb = (nt0 + t) * 16 + 4 * g;
        if (p.part) {  // split K: raw partial sums, the reduce kernel applies the epilogue
            if (ox < p.Wo)
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    const int oy = oy0 + wm * MTW + m;
                    if (oy >= p.Ho) continue;
                    const size_t pl = ((size_t)b * p.Ho + oy) * p.Wo + ox;
                    *reinterpret_cast<c2f4*>(p.part + (((size_t)cls * p.ksplit + ks) * p.B * p.Ho * p.Wo + pl) * p.ncp + cb) = acc[m][t];
                }
            continue;
        }
        if (cb >= p.Cout || ((C2_KO & 16) && p.B > 1000)) continue;
        const C2Chan kc = c2_chan(p, cb, xsc_inv);
        if (ox < p.Wo) {
            const size_t o0 = c2_out_offset(p, bb, od, oy0 + wm * MTW, ox, cls, cb), ostep = c2_row_step(p);
#pragma unroll
            for (int m = 0; m < MTW; ++m)
                if (oy0 + wm * MTW + m < p.Ho) c2_finish(p, o0 + m * ostep, cb, acc[m][t], kc, amax);
        }
    }
    if (p.yamax && !p.part) {
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        if (lane == 0) wmax[wv] = amax;
        __syncthreads();
        if (tid == 0) {
            amax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            raise_absmax_seen(p.yamax, amax, amax_seen);
        }
    }
}

// split K, second kernel: partial sums added in the fixed order ks = 0, 1, ...; then the same epilogue.  One thread = 4 channels.
// (Measured and dropped: no second launch, the workgroup that draws the last ticket of a tile adds the partial sums up.  The
// agent-scope release every workgroup then needs before its ticket writes back its XCD's whole L2: the split layers ran 2-5x slower.)
__global__ void __launch_bounds__(256) c2_reduce_kernel(C2Params p) {
    __shared__ float wmax[4];
    // 32-bit index arithmetic (the launcher checks B Ho Wo Cout / 4 < 2^31): with 64-bit divisions in the decode this pass took
    // 2-3x the time of its reads.  blockIdx.y = parity class of a transposed layer.
    const int npix = p.B * p.Ho * p.Wo;
    const int c4n = (p.Cout + 3) / 4;
    const int cls = blockIdx.y;
    float xsc_inv;
    {
        const unsigned mb = __float_as_uint(*p.xamax);
        const int ex = (int)((mb >> 23) & 0xffu) - 127;
        const int e = max(-125, min(125, ex - 14));
        xsc_inv = __uint_as_float((unsigned)(127 + e) << 23);
    }
    float amax = 0.f;
    // grid-stride: a workgroup ends with at most one atomic
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix * c4n; i += gridDim.x * 256) {
        const int pl = i / c4n;
        const int cb = (i - pl * c4n) * 4;
        // ksplit is a power of two >= 2: eight loads in flight, added in the fixed order ks = 0, 1, ...
        const float* pp = p.part + ((size_t)cls * p.ksplit * npix + pl) * p.ncp + cb;
        const size_t kstr = (size_t)npix * p.ncp;
        c2f4 s = c2f4{0, 0, 0, 0};
        int ks = 0;
        for (; ks + 8 <= p.ksplit; ks += 8) {
            c2f4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const c2f4*>(pp + (size_t)(ks + q) * kstr);
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; ks + 2 <= p.ksplit; ks += 2) {
            const c2f4 v0 = *reinterpret_cast<const c2f4*>(pp + (size_t)ks * kstr), v1 = *reinterpret_cast<const c2f4*>(pp + (size_t)(ks + 1) * kstr);
            s += v0;
            s += v1;
        }
        const int row = pl / p.Wo, ox = pl - row * p.Wo, b = row / p.Ho, oy = row - b * p.Ho;
        const int bb = p.Do == 1 ? b : b / p.Do, od = b - bb * p.Do;
        c2_finish(p, c2_out_offset(p, bb, od, oy, ox, cls, cb), cb, s, c2_chan(p, cb, xsc_inv), amax);
    }
    if (p.yamax) {
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
        __syncthreads();
        if (threadIdx.x == 0) {
            amax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            raise_absmax(p.yamax, amax);
        }
    }
}

// F.interpolate(pred, size = (2h, 2w), mode = "bilinear", align_corners = False) of the decoder (dispnet_decoder.py:131), written
// as C channels of a channel-last slice: x planar (B, C, h, w) -> y[b][oy][ox][0 .. C-1], pixels ys floats apart.  torch's own
// formula and operation order (source index max(0.5 (o + 0.5) - 0.5, 0), the two lambdas, rows blended after columns).
__global__ void __launch_bounds__(256) upsample2x_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, float* yamax, int B, int C, int h,
                                                              int w, int ys) {
    __shared__ float wmax[4];
    const long long n = (long long)B * 4 * h * w;
    float amax = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % (2 * w)), oy = (int)((i / (2 * w)) % (2 * h)), b = (int)(i / (4LL * h * w));
        const float sy = fmaxf(0.5f * ((float)oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * ((float)ox + 0.5f) - 0.5f, 0.f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float ly1 = sy - (float)y0, ly0 = 1.f - ly1, lx1 = sx - (float)x0, lx0 = 1.f - lx1;
        for (int c = 0; c < C; ++c) {
            const float* pl = x + ((size_t)b * C + c) * h * w;
            const float v = ly0 * (lx0 * pl[(size_t)y0 * w + x0] + lx1 * pl[(size_t)y0 * w + x1]) +
                            ly1 * (lx0 * pl[(size_t)y1 * w + x0] + lx1 * pl[(size_t)y1 * w + x1]);
            y[(size_t)i * ys + c] = v;
            amax = fmaxf(amax, finite_abs_or_zero(v));
        }
    }
    if (yamax) {
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
        __syncthreads();
        if (threadIdx.x == 0) {
            amax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            raise_absmax(yamax, amax);
        }
    }
}

// ---- 3-D layers of the regulariser (K4) on the same kernel ------------------------------------------------------------------------
// Conv3d (Cout, Cin, 3, 3, 3), padding 1, stride 1 or 2: the depth taps are chunks (chunk = kd * nchunks_c + c), the in-plane taps the
// 3 x 3 of the 2-D kernel.  ConvTranspose3d (Cin, Cout, 3, 3, 3), stride 2, padding 1, output_padding 1: output voxel 2 i + a takes,
// per dimension, kernel element 1 from input i (a = 0), or elements 2 and 0 from inputs i and i + 1 (a = 1): a 2 x 2 x 2 convolution
// over the input grid (tap t reads input i + t) with 8 Cout output channels (parity class k3 = 4 ad + 2 ay + ax, channel c) at
// k3 * Cout + c, zero weights where a class has no such tap.
__device__ __forceinline__ int c3_deconv_kidx(int a, int t) { return a == 0 ? (t == 0 ? 1 : -1) : (t == 0 ? 2 : 0); }

__global__ void c3_pack_kernel(const float* __restrict__ w, const float* __restrict__ eun, _Float16* __restrict__ packed, int Cin, int Cout,
                               int U8, int nchunks_c, int ntiles, int steps, int transposed, long long total) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int KD = transposed ? 2 : 3, KP = transposed ? 2 : 3;  // depth taps, in-plane taps per dimension
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63), term = (int)((e >> 9) & 1);
    long long r = e >> 10;
    const int step = (int)(r % steps); r /= steps;
    const int chunk = (int)(r % (KD * nchunks_c)); r /= KD * nchunks_c;
    const int nt = (int)r;
    const int kd = chunk / nchunks_c, cc = chunk % nchunks_c;
    const int ce = nt * 16 + (lane & 15), unit = 4 * step + (lane >> 4);
    const int tap = unit / U8, cin = cc * 8 * U8 + 8 * (unit % U8) + j;
    const int ncout = transposed ? 8 * Cout : Cout;
    float v = 0.f;
    if (tap < KP * KP && cin < Cin && ce < ncout) {
        const int ty = tap / KP, tx = tap % KP;
        if (transposed) {
            const int k3 = ce / Cout, c = ce % Cout;
            const int kz = c3_deconv_kidx(k3 >> 2, kd), ky = c3_deconv_kidx((k3 >> 1) & 1, ty), kx = c3_deconv_kidx(k3 & 1, tx);
            if (kz >= 0 && ky >= 0 && kx >= 0) v = w[((((size_t)cin * Cout + c) * 3 + kz) * 3 + ky) * 3 + kx];
        } else {
            v = w[((((size_t)ce * Cin + cin) * 3 + kd) * 3 + ty) * 3 + tx];
        }
        v = v / eun[ce];
    }
    const _Float16 hi = (_Float16)v;
    packed[e] = term == 0 ? hi : (_Float16)(v - (float)hi);
}

// 2^k_c per (class,) channel: max |w| over all of the channel's weights (for the transposed form shared by its 8 classes)
__global__ void c3_wscale_kernel(const float* __restrict__ w, float* __restrict__ eun, int Cin, int Cout, int transposed, int ncout, int cpad) {
    __shared__ float red[256];
    const int ce = blockIdx.x, c = ce % Cout;
    float m = 0.f;
    if (ce < ncout)
        for (int e = threadIdx.x; e < Cin * 27; e += 256) {
            const int ci = e / 27, t = e % 27;
            const float v = fabsf(transposed ? w[((size_t)ci * Cout + c) * 27 + t] : w[((size_t)c * Cin + ci) * 27 + t]);
            m = (v <= 3.4e38f && v > m) ? v : m;
        }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0 && ce < cpad) {
        const int ex = (int)((__float_as_uint(red[0]) >> 23) & 0xffu) - 127;
        const int k = red[0] > 0.f ? max(-100, min(100, ex - 10)) : 0;
        eun[ce] = ldexpf(1.0f, k);
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
struct C2Shape {
    int KH, KW, S, U8, steps;
    bool transposed, nchw3;
};

// the layer kinds that are built.  mode 0: Conv2d, 1: ConvTranspose2d 4 x 4 stride 2 padding 1, 2: Conv2d on a (B, 3, H, W) image
static bool c2_shape(int KH, int KW, int stride, int mode, int cin_pad, C2Shape* s) {
    s->transposed = mode == 1;
    s->nchw3 = mode == 2;
    s->KH = KH; s->KW = KW; s->S = stride;
    if (mode == 1) {
        if (KH != 4 || KW != 4 || stride != 2 || cin_pad % 32) return false;
        s->KH = s->KW = 2; s->S = 1; s->U8 = 4;
    } else if (mode == 2) {
        if (KH != 7 || KW != 7 || stride != 2 || cin_pad != 8) return false;
        s->U8 = 1;
        s->steps = 7;
        return true;
    } else if (KH == 1 && KW == 1 && stride == 1) {
        if (cin_pad % 32) return false;
        s->U8 = 4;
    } else if (KH == 3 && KW == 3 && stride == 1) {
        s->U8 = cin_pad % 32 == 0 ? 4 : 1;
    } else if ((KH == 3 && KW == 3 && stride == 2) || (KH == 5 && KW == 5 && stride == 2)) {
        s->U8 = 1;
    } else {
        return false;
    }
    if (cin_pad % (8 * s->U8)) return false;
    s->steps = (s->KH * s->KW * s->U8 + 3) / 4;
    return true;
}

// output channels per workgroup: the widest tile that does not leave most of it empty
static int c2_bn(int cout) {
    if (const char* e = exp_env("MVD_C2_BN")) {  // experiments library: forced channel tile (tools/sweep_conv2d_plan.py)
        const int v = atoi(e);
        if (v == 16 || v == 32 || v == 64 || v == 128) return v;
    }
    return cout > 64 ? 128 : cout > 32 ? 64 : cout > 16 ? 32 : 16;
}
static int c2_ntiles(int cout) { return (cout + c2_bn(cout) - 1) / c2_bn(cout) * (c2_bn(cout) / 16); }  // 16-channel tiles, whole workgroups
static size_t c2_frag_bytes(const C2Shape& s, int cin_pad, int cout) {
    const size_t nchunks = (size_t)cin_pad / (8 * s.U8);
    return (s.transposed ? 4 : 1) * (size_t)c2_ntiles(cout) * nchunks * s.steps * 2048;
}
static int c2_cpad(int cout) { return (cout + 127) / 128 * 128; }  // eun entries: any BN reads whole float4s

// grid of one layer: output size, channel tile, and over how many workgroups the reduction is split (where the plain grid would
// leave most of the 256 CUs idle and there are chunks to share out)
struct C2Plan {
    int Ho, Wo, tiles_x, tiles_y, bn, nblocks, ncp, ncls, nchunks, ksplit;
    long long tiles;
    size_t part_bytes;
};
static C2Plan c2_plan(const C2Shape& s, int B, int Hi, int Wi, int cin_pad, int cout, int KH, int KW, int stride) {
    C2Plan q{};
    if (s.transposed) { q.Ho = Hi; q.Wo = Wi; q.ncls = 4; }
    else { q.Ho = (Hi + 2 * (KH / 2) - KH) / stride + 1; q.Wo = (Wi + 2 * (KW / 2) - KW) / stride + 1; q.ncls = 1; }
    q.tiles_y = (q.Ho + C2_TH - 1) / C2_TH;
    q.tiles_x = (q.Wo + C2_TW - 1) / C2_TW;
    q.bn = c2_bn(cout);
    q.nblocks = (cout + q.bn - 1) / q.bn;
    q.ncp = q.nblocks * q.bn;
    q.nchunks = cin_pad / (8 * s.U8);
    q.tiles = (long long)q.tiles_x * q.tiles_y * B * q.nblocks * q.ncls;
    q.ksplit = 1;  // a grid of 128+ workgroups runs as it is: the second pass would cost more than the idle CUs
    if (q.tiles < 128)
        while (q.tiles * q.ksplit < 320 && q.ksplit < 32 && q.nchunks / (q.ksplit * 2) >= 2) q.ksplit *= 2;
    if (const char* e = exp_env("MVD_C2_KSPLIT")) {  // experiments library: forced split of the reduction
        const int v = atoi(e);
        if (v >= 1 && v <= 32 && (v & (v - 1)) == 0 && v <= q.nchunks) q.ksplit = v;
    }
    q.part_bytes = q.ksplit > 1 ? (size_t)q.ksplit * q.ncls * B * q.Ho * q.Wo * q.ncp * sizeof(float) : 0;
    return q;
}

template <int KH, int KW, int S, int U8, int WM, int NTW, bool NCHW3>
static int c2_launch(const C2Params& p, long long nblk, hipStream_t st) {
    using G = C2Geom<KH, KW, S, U8, NCHW3>;
    auto kern = conv2d_split_kernel<KH, KW, S, U8, WM, NTW, NCHW3>;
    constexpr int lds = 2 * G::PLANE + 16;
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return launch_status("conv2d_split: LDS attribute");
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, p);
    return launch_status("conv2d_split");
}

// tile of output channels per workgroup: bn = 16 WN NTW
template <int KH, int KW, int S, int U8, bool NCHW3>
static int c2_launch_bn(const C2Params& p, int bn, long long nblk, hipStream_t st) {
    switch (bn) {
        case 128: return c2_launch<KH, KW, S, U8, 1, 2, NCHW3>(p, nblk, st);
        case 64: return c2_launch<KH, KW, S, U8, 1, 1, NCHW3>(p, nblk, st);
        case 32: return c2_launch<KH, KW, S, U8, 2, 1, NCHW3>(p, nblk, st);
        case 16: return c2_launch<KH, KW, S, U8, 4, 1, NCHW3>(p, nblk, st);
    }
    return MVD_ERR_INVALID_ARG;
}

static int c2_dispatch(const C2Shape& s, const C2Params& p, int bn, long long nblk, hipStream_t st) {
    if (s.nchw3) return c2_launch_bn<7, 7, 2, 1, true>(p, bn, nblk, st);
    if (s.KH == 1) return c2_launch_bn<1, 1, 1, 4, false>(p, bn, nblk, st);
    if (s.KH == 2) return s.U8 == 4 ? c2_launch_bn<2, 2, 1, 4, false>(p, bn, nblk, st) : c2_launch_bn<2, 2, 1, 2, false>(p, bn, nblk, st);
    if (s.KH == 5) return c2_launch_bn<5, 5, 2, 1, false>(p, bn, nblk, st);
    if (s.S == 2) return c2_launch_bn<3, 3, 2, 1, false>(p, bn, nblk, st);
    if (s.U8 == 4) return c2_launch_bn<3, 3, 1, 4, false>(p, bn, nblk, st);
    return c2_launch_bn<3, 3, 1, 1, false>(p, bn, nblk, st);
}

}  // namespace mvd

extern "C" {

size_t mvd_conv2d_split_packed_weight_bytes(int Cin_pad, int Cout, int KH, int KW, int stride, int mode) {
    mvd::C2Shape s;
    if (Cin_pad <= 0 || Cout <= 0 || !mvd::c2_shape(KH, KW, stride, mode, Cin_pad, &s)) return 0;
    return mvd::c2_frag_bytes(s, Cin_pad, Cout) + (size_t)mvd::c2_cpad(Cout) * sizeof(float);
}

int mvd_pack_conv2d_weights_split(const float* w, int Cin, int Cin_pad, int Cout, int KH, int KW, int stride, int mode, void* packed,
                                  mvd_stream_t stream) {
    MVD_REQUIRE(w && packed, "pack_conv2d_weights_split: NULL argument");
    mvd::C2Shape s;
    MVD_REQUIRE(Cin > 0 && Cin <= Cin_pad && Cout > 0 && mvd::c2_shape(KH, KW, stride, mode, Cin_pad, &s),
                "pack_conv2d_weights_split: layer %dx%d stride %d mode %d with %d (padded %d) -> %d channels is not built", KH, KW, stride, mode,
                Cin, Cin_pad, Cout);
    hipStream_t st = (hipStream_t)stream;
    const size_t fb = mvd::c2_frag_bytes(s, Cin_pad, Cout);
    float* eun = reinterpret_cast<float*>(static_cast<char*>(packed) + fb);
    const int cpad = mvd::c2_cpad(Cout);
    hipLaunchKernelGGL(mvd::c2_wscale_kernel, dim3(cpad), dim3(256), 0, st, w, eun, Cin, Cout, KH * KW, s.transposed ? 1 : 0, cpad);
    const long long total = (long long)(fb / 2);
    hipLaunchKernelGGL(mvd::c2_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, eun, (_Float16*)packed, Cin, Cout, s.KH,
                       s.KW, s.U8, Cin_pad / (8 * s.U8), mvd::c2_ntiles(Cout), s.steps, s.transposed ? 1 : s.nchw3 ? 2 : 0, total);
    return mvd::launch_status("pack_conv2d_weights_split");
}

size_t mvd_conv2d_split_workspace_bytes(int B, int Hi, int Wi, int Cin_pad, int Cout, int KH, int KW, int stride, int mode) {
    mvd::C2Shape s;
    if (B <= 0 || Hi <= 0 || Wi <= 0 || Cin_pad <= 0 || Cout <= 0 || !mvd::c2_shape(KH, KW, stride, mode, Cin_pad, &s)) return 0;
    return mvd::c2_plan(s, B, Hi, Wi, Cin_pad, Cout, KH, KW, stride).part_bytes;
}

int mvd_conv2d_split_f32(const float* x, const float* x_absmax, const void* packed_w, const float* bias, float* y, float* y_absmax, int B,
                         int Hi, int Wi, int Cin_pad, int x_pixel_stride, int Cout, int y_pixel_stride, long long y_row_stride,
                         long long y_image_stride, long long y_channel_stride, int KH, int KW, int stride, int mode, int act, float slope,
                         void* workspace, size_t workspace_bytes, mvd_stream_t stream) {
    MVD_REQUIRE(x && x_absmax && packed_w && y, "conv2d_split: NULL argument");
    MVD_REQUIRE(B > 0 && Hi > 0 && Wi > 0, "conv2d_split: non-positive dimension");
    mvd::C2Shape s;
    MVD_REQUIRE(Cin_pad > 0 && Cout > 0 && mvd::c2_shape(KH, KW, stride, mode, Cin_pad, &s),
                "conv2d_split: layer %dx%d stride %d mode %d with %d -> %d channels is not built", KH, KW, stride, mode, Cin_pad, Cout);
    MVD_REQUIRE(mode == 2 || (x_pixel_stride >= Cin_pad && x_pixel_stride % 4 == 0), "conv2d_split: input pixel stride %d (channels %d)",
                x_pixel_stride, Cin_pad);
    if (y_channel_stride <= 0) y_channel_stride = 1;
    MVD_REQUIRE(y_channel_stride != 1 || y_pixel_stride >= Cout, "conv2d_split: output pixel stride %d below %d channels", y_pixel_stride, Cout);
    MVD_REQUIRE(y_pixel_stride >= 1 && y_row_stride >= 0 && y_image_stride >= 0, "conv2d_split: negative output stride");
    MVD_REQUIRE(act >= 0 && act <= 2, "conv2d_split: act=%d unknown", act);
    MVD_REQUIRE(mode == 2 || (((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0), "conv2d_split: x and y must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    mvd::C2Params p{};
    p.x = x; p.xamax = x_absmax; p.bias = bias; p.y = y; p.yamax = y_absmax;
    p.B = B; p.Hi = Hi; p.Wi = Wi; p.xs = x_pixel_stride; p.nchunks = Cin_pad / (8 * s.U8);
    p.Di = p.Do = p.Dy = p.KD = p.SD = 1; p.pad_d = 0; p.nchunks_c = p.nchunks; p.sub3d = 0; p.Cr = Cout;
    p.Cout = Cout; p.ys = y_pixel_stride; p.act = act; p.slope = slope;
    const size_t fb = mvd::c2_frag_bytes(s, Cin_pad, Cout);
    p.eun = reinterpret_cast<const float*>(static_cast<const char*>(packed_w) + fb);
    const mvd::C2Plan q = mvd::c2_plan(s, B, Hi, Wi, Cin_pad, Cout, KH, KW, stride);
    p.Ho = q.Ho; p.Wo = q.Wo; p.ncls = q.ncls;
    if (s.transposed) {
        p.Hy = 2 * Hi; p.Wy = 2 * Wi; p.oy_mul = p.ox_mul = 2; p.pad_y = p.pad_x = 1;
    } else {
        p.Hy = p.Ho; p.Wy = p.Wo; p.oy_mul = p.ox_mul = 1; p.pad_y = KH / 2; p.pad_x = KW / 2;
    }
    MVD_REQUIRE(y_channel_stride != 1 || Cout < 4 || (y_pixel_stride % 4 == 0 && y_row_stride % 4 == 0 && y_image_stride % 4 == 0),
                "conv2d_split: channel-last output strides must be multiples of 4 floats");
    p.ycs = y_channel_stride;
    p.ys_row = y_row_stride > 0 ? y_row_stride : (long long)p.Wy * y_pixel_stride;
    p.ys_img = y_image_stride > 0 ? y_image_stride : (long long)p.Hy * p.ys_row;
    p.wpk = static_cast<const char*>(packed_w);
    p.cls_bytes = (long long)(fb / p.ncls);
    p.tiles_y = q.tiles_y; p.tiles_x = q.tiles_x;
    const int bn = q.bn;
    p.nblocks = q.nblocks;
    p.ncp = q.ncp;
    const long long tiles = q.tiles;
    int ksplit = q.ksplit;
    if (ksplit > 1 && (!workspace || workspace_bytes < q.part_bytes)) ksplit = 1;  // no workspace: the plain grid
    p.ksplit = ksplit;
    p.chunks_per_split = (p.nchunks + ksplit - 1) / ksplit;
    p.part = ksplit > 1 ? static_cast<float*>(workspace) : nullptr;
    const long long nblk = tiles * ksplit;
    MVD_REQUIRE(nblk <= 0x7fffffffLL, "conv2d_split: %lld workgroups exceed the grid limit", nblk);
    int rc = mvd::c2_dispatch(s, p, bn, nblk, st);
    if (rc != MVD_OK || ksplit == 1) return rc;
    const long long n = (long long)B * p.Ho * p.Wo * ((Cout + 3) / 4);
    MVD_REQUIRE(n < 0x7fffffffLL, "conv2d_split: split reduction over %lld outputs", n);
    hipLaunchKernelGGL(mvd::c2_reduce_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 1024), (unsigned)p.ncls), dim3(256), 0, st, p);
    return mvd::launch_status("conv2d_split: reduce");
}

int mvd_upsample2x_nhwc_f32(const float* x, float* y, float* y_absmax, int B, int C, int h, int w, int y_pixel_stride, mvd_stream_t stream) {
    MVD_REQUIRE(x && y, "upsample2x_nhwc: NULL argument");
    MVD_REQUIRE(B > 0 && C > 0 && h > 0 && w > 0 && y_pixel_stride >= C, "upsample2x_nhwc: bad dimension");
    const long long n = (long long)B * 4 * h * w;
    hipLaunchKernelGGL(mvd::upsample2x_nhwc_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 1024)), dim3(256), 0, (hipStream_t)stream, x, y,
                       y_absmax, B, C, h, w, y_pixel_stride);
    return mvd::launch_status("upsample2x_nhwc");
}

/* ---- 3-D layers ---- */
}  // extern "C"

namespace mvd {
// mode: MVD_CONV3D_STRIDE1 / MVD_CONV3D_STRIDE2 / MVD_DECONV3D_STRIDE2 (mvd.h)
static bool c3_shape(int Cin, int Cout, int mode, C2Shape* s, int* ncout) {
    if (Cin <= 0 || Cin % 8 || Cout <= 0 || Cout % 4 || mode < 0 || mode > 2) return false;
    s->transposed = false; s->nchw3 = false;
    if (mode == MVD_DECONV3D_STRIDE2) {
        if (Cin % 16) return false;
        s->KH = s->KW = 2; s->S = 1; s->U8 = Cin % 32 == 0 ? 4 : 2;
        *ncout = 8 * Cout;
    } else {
        s->KH = s->KW = 3; s->S = mode == MVD_CONV3D_STRIDE2 ? 2 : 1;
        s->U8 = (s->S == 1 && Cin % 32 == 0) ? 4 : 1;
        *ncout = Cout;
    }
    s->steps = (s->KH * s->KW * s->U8 + 3) / 4;
    return true;
}
static size_t c3_frag_bytes(const C2Shape& s, int Cin, int ncout, int mode) {
    const size_t KD = mode == MVD_DECONV3D_STRIDE2 ? 2 : 3;
    return (size_t)c2_ntiles(ncout) * KD * (Cin / (8 * s.U8)) * s.steps * 2048;
}
}  // namespace mvd

extern "C" {

size_t mvd_conv3d_igemm_packed_weight_bytes(int Cin, int Cout, int mode) {
    mvd::C2Shape s;
    int ncout;
    if (!mvd::c3_shape(Cin, Cout, mode, &s, &ncout)) return 0;
    return mvd::c3_frag_bytes(s, Cin, ncout, mode) + (size_t)mvd::c2_cpad(ncout) * sizeof(float);
}

int mvd_pack_conv3d_weights_igemm(const float* w, int Cin, int Cout, int mode, void* packed, mvd_stream_t stream) {
    MVD_REQUIRE(w && packed, "pack_conv3d_weights_igemm: NULL argument");
    mvd::C2Shape s;
    int ncout;
    MVD_REQUIRE(mvd::c3_shape(Cin, Cout, mode, &s, &ncout), "pack_conv3d_weights_igemm: %d -> %d channels, mode %d is not built", Cin, Cout, mode);
    hipStream_t st = (hipStream_t)stream;
    const size_t fb = mvd::c3_frag_bytes(s, Cin, ncout, mode);
    float* eun = reinterpret_cast<float*>(static_cast<char*>(packed) + fb);
    const int cpad = mvd::c2_cpad(ncout), tr = mode == MVD_DECONV3D_STRIDE2 ? 1 : 0;
    hipLaunchKernelGGL(mvd::c3_wscale_kernel, dim3(cpad), dim3(256), 0, st, w, eun, Cin, Cout, tr, ncout, cpad);
    const long long total = (long long)(fb / 2);
    hipLaunchKernelGGL(mvd::c3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, eun, (_Float16*)packed, Cin, Cout, s.U8,
                       Cin / (8 * s.U8), mvd::c2_ntiles(ncout), s.steps, tr, total);
    return mvd::launch_status("pack_conv3d_weights_igemm");
}

static mvd::C2Plan c3_plan(const mvd::C2Shape& s, int B, int Di, int Hi, int Wi, int Cin, int ncout, int mode, int* Do) {
    mvd::C2Plan q{};
    const int sd = mode == MVD_CONV3D_STRIDE2 ? 2 : 1;
    *Do = Di / sd; q.Ho = Hi / sd; q.Wo = Wi / sd; q.ncls = 1;
    q.tiles_y = (q.Ho + mvd::C2_TH - 1) / mvd::C2_TH;
    q.tiles_x = (q.Wo + mvd::C2_TW - 1) / mvd::C2_TW;
    q.bn = mvd::c2_bn(ncout);
    q.nblocks = (ncout + q.bn - 1) / q.bn;
    q.ncp = q.nblocks * q.bn;
    q.nchunks = (mode == MVD_DECONV3D_STRIDE2 ? 2 : 3) * (Cin / (8 * s.U8));
    q.tiles = (long long)q.tiles_x * q.tiles_y * B * *Do * q.nblocks;
    q.ksplit = 1;
    if (q.tiles < 128)
        while (q.tiles * q.ksplit < 320 && q.ksplit < 32 && q.nchunks / (q.ksplit * 2) >= 2) q.ksplit *= 2;
    q.part_bytes = q.ksplit > 1 ? (size_t)q.ksplit * B * *Do * q.Ho * q.Wo * q.ncp * sizeof(float) : 0;
    return q;
}

size_t mvd_conv3d_igemm_workspace_bytes(int B, int Di, int Hi, int Wi, int Cin, int Cout, int mode) {
    mvd::C2Shape s;
    int ncout, Do;
    if (B <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || !mvd::c3_shape(Cin, Cout, mode, &s, &ncout)) return 0;
    return c3_plan(s, B, Di, Hi, Wi, Cin, ncout, mode, &Do).part_bytes;
}

int mvd_conv3d_bn_relu_igemm_f32(const float* x, const float* x_absmax, const void* packed_w, const float* scale, const float* shift,
                                 const float* skip, float* y, float* y_absmax, int B, int Di, int Hi, int Wi, int Cin, int Cout, int mode,
                                 int relu, void* workspace, size_t workspace_bytes, mvd_stream_t stream) {
    MVD_REQUIRE(x && x_absmax && packed_w && scale && shift && y, "conv3d_igemm: NULL argument");
    MVD_REQUIRE(B > 0 && Di > 0 && Hi > 0 && Wi > 0, "conv3d_igemm: non-positive dimension");
    mvd::C2Shape s;
    int ncout;
    MVD_REQUIRE(mvd::c3_shape(Cin, Cout, mode, &s, &ncout), "conv3d_igemm: %d -> %d channels, mode %d is not built", Cin, Cout, mode);
    MVD_REQUIRE(mode != MVD_CONV3D_STRIDE2 || (Di % 2 == 0 && Hi % 2 == 0 && Wi % 2 == 0), "conv3d_igemm stride 2: odd input dims %dx%dx%d", Di, Hi, Wi);
    MVD_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)skip) & 15) == 0, "conv3d_igemm: x, y and skip must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    int Do;
    const mvd::C2Plan q = c3_plan(s, B, Di, Hi, Wi, Cin, ncout, mode, &Do);
    const bool tr = mode == MVD_DECONV3D_STRIDE2;
    mvd::C2Params p{};
    p.x = x; p.xamax = x_absmax; p.bias = shift; p.scale = scale; p.skip = skip; p.y = y; p.yamax = y_absmax;
    p.wpk = static_cast<const char*>(packed_w);
    p.eun = reinterpret_cast<const float*>(p.wpk + mvd::c3_frag_bytes(s, Cin, ncout, mode));
    p.B = B * Do; p.Hi = Hi; p.Wi = Wi; p.xs = Cin; p.nchunks = q.nchunks;
    p.Di = Di; p.Do = Do; p.KD = tr ? 2 : 3; p.SD = mode == MVD_CONV3D_STRIDE2 ? 2 : 1; p.pad_d = tr ? 0 : 1; p.nchunks_c = Cin / (8 * s.U8);
    p.sub3d = tr ? 1 : 0; p.Cr = Cout; p.Dy = tr ? 2 * Di : Do;
    p.Ho = q.Ho; p.Wo = q.Wo; p.Hy = tr ? 2 * Hi : q.Ho; p.Wy = tr ? 2 * Wi : q.Wo;
    p.oy_mul = p.ox_mul = 1; p.pad_y = p.pad_x = tr ? 0 : 1;
    p.Cout = ncout; p.ncp = q.ncp; p.ys = Cout; p.ycs = 1;
    p.ys_row = (long long)p.Wy * Cout; p.ys_img = (long long)p.Hy * p.ys_row;
    p.tiles_x = q.tiles_x; p.tiles_y = q.tiles_y; p.nblocks = q.nblocks; p.ncls = 1; p.cls_bytes = 0;
    p.act = relu ? 2 : 0; p.slope = 0.f;
    int ksplit = q.ksplit;
    if (ksplit > 1 && (!workspace || workspace_bytes < q.part_bytes)) ksplit = 1;
    p.ksplit = ksplit;
    p.chunks_per_split = (p.nchunks + ksplit - 1) / ksplit;
    p.part = ksplit > 1 ? static_cast<float*>(workspace) : nullptr;
    const long long nblk = q.tiles * ksplit;
    MVD_REQUIRE(nblk <= 0x7fffffffLL, "conv3d_igemm: %lld workgroups exceed the grid limit", nblk);
    int rc = mvd::c2_dispatch(s, p, q.bn, nblk, st);
    if (rc != MVD_OK || ksplit == 1) return rc;
    const long long n = (long long)p.B * p.Ho * p.Wo * ((p.Cout + 3) / 4);
    MVD_REQUIRE(n < 0x7fffffffLL, "conv3d_igemm: split reduction over %lld outputs", n);
    hipLaunchKernelGGL(mvd::c2_reduce_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 1024)), dim3(256), 0, st, p);
    return mvd::launch_status("conv3d_igemm: reduce");
}
}
