// K4 first layer with fp32 operands split into two fp16 terms each, range-safe (the default first layer of the regulariser).
//
// conv0 of CostRegNet (rmvd/models/blocks/mvsnet_components.py:78; 32 -> 8, 3x3x3, stride 1, padding 1) is 44 % of the
// headline step and is capped by the fp32 matrix rate (v_mfma_f32_16x16x4_f32 = the fp32 VALU rate, 1/16 of fp16 MFMA).
// Here every fp32 activation a and weight w is split exactly into
//     a = a_hi + 2^-11 a_lo,   a_hi = fp16(a),  a_lo = fp16((a - a_hi) * 2^11)      (same for w)
// (a - a_hi is exact in fp32, the scaling by 2^11 keeps a_lo in fp16's normal range) and the product is evaluated as
//     a w  ~=  a_hi w_hi + 2^-11 (a_hi w_lo + a_lo w_hi)                  dropped: 2^-22 a_lo w_lo
// on v_mfma_f32_16x16x32_f16 with fp32 accumulation: relative error per product <= ~3 * 2^-22 (fp32 itself: 2^-24), measured
// against the fp32-MFMA kernel in tests/test_hip_f16.py.  A = activations (16 voxels), B = weights with the 16 columns
// [w_hi of the 8 couts | w_lo of the 8 couts]: one MFMA with A = a_hi yields both a_hi w_hi and a_hi w_lo (all 16 columns
// useful), a second with A = a_lo yields a_lo w_hi (half useful).  2 MFMAs of 16 cycles per (tap, 16 voxels) replace 8
// fp32 MFMAs of 32 cycles: the layer turns from matrix-bound into an LDS / memory pass.
//
// Range: fp16 spans 2^-24 .. 65504, fp32 operands do not.  Both operands are therefore multiplied by exact powers of two
// before the split and the result by the inverse afterwards (block floating point, nothing is rounded by the scaling):
//   * activations by 2^-e, e = exponent(max |x| over the whole input) - 14, so that the largest magnitude lands in
//     [2^14, 2^15).  The caller supplies max |x| in device memory (mvd_absmax_f32, or the by-product of
//     mvd_warp_variance_absmax_f32 which costs nothing extra); any upper bound is safe, a tight one is most precise.
//     A value v is then represented to |error| <= max(2^-22 |v|, 2^-50 max|x|): fp32-grade for everything within 2^-26 of the
//     largest magnitude, and an ABSOLUTE error below fp32's own rounding of the larger terms for what lies underneath.
//   * each output channel's weights by 2^-k_c, k_c = exponent(max |w_c|) - 10 (computed when the weights are packed).
// The epilogue multiplies by 2^(e + k_c) before the batch-norm affine.  Inputs of any uniform magnitude (1e-30 .. 1e30,
// denormals included) give the same relative accuracy; inf / NaN inputs give inf / NaN outputs where they reach, as on fp32.
//
// Workgroup (4 waves) = 4 x 32 tile marching through TD + 2 input planes (see the kernel); the fp32 volume is converted to
// the two fp16 terms while it is staged (global -> registers -> LDS), all 27 weight fragments stay in registers.
#include "mvd_common.h"

// Knock-out / tuning macros for tools/ko_conv0_split.sh (timing only: the results of a knocked-out build are wrong).
#ifndef SPLIT_KO
#define SPLIT_KO 0  // 1 no reloads, 2 no stores, 4 no split + LDS writes, 8 no barrier, 16 no MFMAs, 32 no fragment reads
#endif
#ifndef SPLIT_DEPTH
#define SPLIT_DEPTH 4
#endif

namespace mvd {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float sf32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int su32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int su32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

constexpr int S_TH = 4, S_TW = 32, S_ROWS = S_TH + 2;
// geometry by input channel count.  CIN = 32: one voxel's channels are one MFMA K (64 bytes per term), three fragments per
// (kd, kh) — one per kw.  CIN = 16: a 64-byte fragment spans TWO x-adjacent voxels, so the kw taps go in pairs (0,1), (2, zero
// weights): two fragments per (kd, kh); the staged tile gets a 35th, always-zero column for the pad tap to read.
template <int CIN>
struct SplitGeom {
    static constexpr int CH8 = CIN / 8;                         // 8-channel chunks per voxel
    static constexpr int VB = CIN * 2;                          // bytes per voxel and term
    static constexpr int KWG = CIN == 32 ? 3 : 2;               // fragments per (kd, kh)
    static constexpr int NT = 9 * KWG;                          // weight fragments per cout block
    static constexpr int COLS = CIN == 32 ? S_TW + 2 : S_TW + 3;
    static constexpr int HALF_BYTES = S_ROWS * COLS * VB;       // hi (or lo) part of one staged plane
    static constexpr int PLANE_BYTES = 2 * HALF_BYTES;
    static constexpr int ITEMS = S_ROWS * COLS * CH8;           // (voxel, 8-channel chunk) items per plane
    static constexpr int NLOAD = (ITEMS + 255) / 256;
    static constexpr int NF = 4 * KWG * 2;                      // fragments per pass: input rows x kw groups x terms
    static_assert(NF - 8 == 4 * NLOAD, "pass B interleaves 8 epilogue groups and 4 staging groups per item");
};

// per output channel: 2^k_c with k_c = exponent(max |w_c|) - 10 (0 for an all-zero channel), appended to the packed buffer
__global__ void conv0_split_wscale_kernel(const float* __restrict__ w, float* __restrict__ wscale, int cin) {
    __shared__ float red[256];
    const int c = blockIdx.x;
    float m = 0.f;
    for (int e = threadIdx.x; e < cin * 27; e += 256) {
        const float v = fabsf(w[(size_t)c * cin * 27 + e]);
        m = (v <= 3.4e38f && v > m) ? v : m;  // finite values only
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int ex = (int)((__float_as_uint(red[0]) >> 23) & 0xffu) - 127;
        const int k = red[0] > 0.f ? max(-100, min(100, ex - 10)) : 0;
        wscale[c] = ldexpf(1.0f, k);
    }
}

// w (Cout, CIN, 3, 3, 3) fp32 -> [cout block Cout/8][fragment NT][lane 64][8 halves]: lane l = column l%16 (0..7: w_hi of cout
// 8 cb + l%16, 8..15: w_lo of cout 8 cb + l%16 - 8), K values 8*(l/16) .. +7 of the fragment: CIN = 32: fragment = tap, K = cin;
// CIN = 16: fragment = (kd, kh, kw pair g), K = 16 * (kw - 2g) + cin, kw = 3 is the zero pad.  Weights divided by the
// channel's 2^k_c first (exact).
template <int CIN>
__global__ void pack_conv0_split_kernel(const float* __restrict__ w, const float* __restrict__ wscale, _Float16* __restrict__ packed, int ncb) {
    using G = SplitGeom<CIN>;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= ncb * G::NT * 64 * 8) return;
    const int j = e & 7, lane = (e >> 3) & 63, t = (e >> 9) % G::NT, cb = (e >> 9) / G::NT;
    const int col = lane & 15, k = 8 * (lane >> 4) + j, cout = cb * 8 + (col & 7);
    int tap, cin;
    if constexpr (CIN == 32) { tap = t; cin = k; }
    else { const int kw = 2 * (t % 2) + k / 16; tap = kw < 3 ? (t / 2) * 3 + kw : -1; cin = k % 16; }
    const float v = tap >= 0 ? w[((size_t)cout * CIN + cin) * 27 + tap] / wscale[cout] : 0.0f;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)((v - (float)hi) * 2048.0f);
    packed[e] = col < 8 ? hi : lo;
}

struct SplitParams {
    const float* x;       // (B, D, h, w, 32) fp32
    const char* wpk;
    const float* scale;
    const float* shift;
    const float* absmax;  // max |x| (device, one float): sets the activations' power-of-two scale
    float* y;             // (B, D, h, w, Cout) fp32
    float* yamax;         // optional (device, one float, zeroed by the launcher): max |y| over the finite outputs
    int B, D, h, w, relu;
    int cout, ncb;        // output channels (a multiple of 8) and Cout / 8: one workgroup computes 8 of them
    int tiles_x, tiles_y, dgroups, td, tiles_per_xcd;
};

__device__ __forceinline__ unsigned pack_h2(_Float16 a, _Float16 b) {
    const h16x2 v = {a, b};
    return __builtin_bit_cast(unsigned, v);
}

// Plane-stationary march.  One input plane z is staged per step and every activation fragment read from LDS is used for all
// the outputs it feeds in d and y: the three output planes z-1, z, z+1 (taps kd = 2, 1, 0) and the wave's two output rows
// (taps kh = rr - row).  24 fragment reads feed the 108 MFMAs of a step (4.5 per read); with one read per MFMA (the first
// version, profiles/r02_conv0_split_pmc.txt) the four SIMDs of a CU asked the LDS for 32 clocks of ds_read_b128 per 16
// clocks of MFMA and the kernel sat on the LDS port at 35 % matrix utilisation.  Three output planes of accumulators are in
// flight per wave (48 VGPRs); the LDS holds two plane slots (the one being read, the one being written).
// wave wv: columns 16 (wv & 1) .. +15, output rows 2 (wv >> 1), +1 of the 4 x 32 tile.
template <int CIN, bool YAMAX = false>  // YAMAX: also max |y| (a variant of its own: the extra epilogue work costs the plain one 7 %)
__global__ void __launch_bounds__(256, 2) conv0_split_kernel(SplitParams p) {
    using G = SplitGeom<CIN>;
    constexpr int S_COLS = G::COLS, S_HALF_BYTES = G::HALF_BYTES, S_PLANE_BYTES = G::PLANE_BYTES, S_ITEMS = G::ITEMS, S_NLOAD = G::NLOAD;
    constexpr int KWG = G::KWG, NT = G::NT;
    extern __shared__ __attribute__((aligned(16))) char ring[];  // 2 planes x (hi part | lo part)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int D = p.D, h = p.h, w = p.w;

    // XCD-aware decode (blocks b and b + 8 share an XCD and its L2): each XCD owns a contiguous run of (row-major) tiles, so
    // the halo rows and columns that neighbouring tiles share are re-read from ONE L2 instead of from HBM by eight
    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    const int cb = j % p.ncb; j /= p.ncb;  // the cout blocks of a tile run next to each other: they read the same input through one L2
    const int t_in = j % p.tiles_per_xcd; j /= p.tiles_per_xcd;
    const int dg = j % p.dgroups;
    const int b = j / p.dgroups;
    const int tile = xcd * p.tiles_per_xcd + t_in;
    if (tile >= p.tiles_x * p.tiles_y) return;  // block-uniform
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const int x0 = tx * S_TW, y0 = ty * S_TH;
    const int dz0 = dg * p.td, dz1 = min(dz0 + p.td, D);

    h16x8 wf[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wf[t] = *reinterpret_cast<const h16x8*>(p.wpk + (((size_t)cb * NT + t) * 64 + lane) * 16);

    // staging map: item e = tid + 256 k -> (row, col, 8-channel chunk).  Everything below is branch-free: a voxel outside the
    // plane (or a plane outside the volume) is an out-of-range buffer offset (reads 0), the 208 idle items of the last round
    // write to a dump row behind the two slots.  With branches the compiler cannot count what is outstanding and falls back to
    // vmcnt(0), which also waits for the stores.
    constexpr unsigned OOB = 0x80000000u;
    unsigned gob[S_NLOAD];
    int loff[S_NLOAD];
#pragma unroll
    for (int k = 0; k < S_NLOAD; ++k) {
        const int e = tid + 256 * k;
        const int vox = e / G::CH8, ch = e % G::CH8;
        const int r = vox / S_COLS, c = vox - r * S_COLS;
        const int gy = y0 - 1 + r, gx = x0 - 1 + c;
        const bool in = e < S_ITEMS && c < S_TW + 2 && gy >= 0 && gy < h && gx >= 0 && gx < w;  // (the 35th column of CIN = 16 stays zero)
        gob[k] = in ? (unsigned)(((gy * w + gx) * CIN + ch * 8) * 4) : OOB;                 // bytes inside a plane
        if constexpr (CIN == 32) loff[k] = e < S_ITEMS ? (vox * 64 + ((ch ^ ((c >> 1) & 3)) * 16)) : -1;  // swizzled as in conv0_f16
        else loff[k] = e < S_ITEMS ? e * 16 : -1;  // 32 bytes per voxel: fragment reads are conflict-free as they stand
    }
    const size_t plane_f = (size_t)h * w * CIN;
    const float* xb = p.x + (size_t)b * D * plane_f;
    su32x4 pre[S_NLOAD][2];
    auto fetch = [&](int d) {
        const bool din = d >= 0 && d < D && d <= dz1;  // block-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(xb + (din ? (size_t)d * plane_f : 0)), 0, din ? (int)(plane_f * 4) : 0, 0x00020000);
#pragma unroll
        for (int k = 0; k < S_NLOAD; ++k) {
            pre[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, gob[k], 0, 0);
            pre[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, gob[k] + 16u, 0, 0);
        }
    };
    char* const dump = ring + 2 * S_PLANE_BYTES + tid * 16;
    // split two fp32 values into the packed fp16 pairs (hi, lo * 2^11): hi = fp16(a), t = a - hi (exact, mixed-precision fma
    // straight from the packed half), lo = fp16(t * 2^11) written into its half.  5 instructions per pair; the compiler's own
    // sequence for the same arithmetic is 10 (it converts every hi twice and back).
    // activation scale 2^-e, e = exponent(max |x|) - 14 clamped to what a normal fp32 power of two can express
    float k2048 = 2048.0f, xs, xs_inv;
    {
        const unsigned mb = __builtin_amdgcn_readfirstlane((int)__float_as_uint(*p.absmax));
        const int ex = (int)((mb >> 23) & 0xffu) - 127;
        const int e = max(-125, min(125, ex - 14));
        xs = __uint_as_float((unsigned)(127 - e) << 23);
        xs_inv = __uint_as_float((unsigned)(127 + e) << 23);
    }
    auto split2 = [k2048, xs](float a0, float a1, unsigned& hi, unsigned& lo) {
        unsigned hp, lp;
        float t0, t1;
        // hi = fp16(a * 2^-e) (one rounding: the scaling is exact), t = a * 2^-e - hi (exact), lo = fp16(t * 2^11)
        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(hp) : "v"(a0), "s"(xs));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(hp) : "v"(a1), "s"(xs));
        asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(t0) : "v"(a0), "s"(xs), "v"(hp));
        asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(t1) : "v"(a1), "s"(xs), "v"(hp));
        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(lp) : "v"(t0), "s"(k2048));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(lp) : "v"(t1), "s"(k2048));
        hi = hp;
        lo = lp;
    };
    auto stash = [&](int d) {  // prologue only: the loop below stages inside the MFMA stream
        char* slot = ring + (d & 1) * S_PLANE_BYTES;
#pragma unroll
        for (int k = 0; k < S_NLOAD; ++k) {
            su32x4 hv, lv;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const sf32x4 v = __builtin_bit_cast(sf32x4, pre[k][q >> 1]);
                unsigned hi, lo;
                split2(v[(q & 1) * 2], v[(q & 1) * 2 + 1], hi, lo);
                hv[q] = hi; lv[q] = lo;
            }
            char* dst = loff[k] >= 0 ? slot + loff[k] : dump;
            *reinterpret_cast<su32x4*>(dst) = hv;
            *reinterpret_cast<su32x4*>(loff[k] >= 0 ? dst + S_HALF_BYTES : dst) = lv;
        }
    };

    // epilogue constants of the lanes that end up with a result: column (cout) l%16 < 8
    const int col = lane & 15;
    float esc_ = p.scale[cb * 8 + (col & 7)], esh_ = p.shift[cb * 8 + (col & 7)];
    float eun_ = reinterpret_cast<const float*>(p.wpk + (size_t)p.ncb * NT * 64 * 16)[cb * 8 + (col & 7)] * xs_inv;  // 2^(k_c + e): undoes both scalings
    const float floor_ = p.relu ? 0.f : -__builtin_inff();
    const int xh = wv & 1, rp = wv >> 1;

    int fragk[KWG];
#pragma unroll
    for (int kw = 0; kw < KWG; ++kw) {
        if constexpr (CIN == 32) {
            const int c = 16 * xh + (lane & 15) + kw;
            fragk[kw] = (2 * rp * S_COLS + c) * 64 + (((lane >> 4) ^ ((c >> 1) & 3)) * 16);
        } else {  // kw pair: K slices 0,1 = voxel c, slices 2,3 = voxel c + 1 (32 bytes further)
            const int c = 16 * xh + (lane & 15) + 2 * kw;
            fragk[kw] = (2 * rp * S_COLS + c) * 32 + (lane >> 4) * 16;
        }
    }

    // acc[plane slot][row][term]: slot 0 = output plane z-1 (finishes in this step), 1 = plane z, 2 = plane z+1 (starts)
    sf32x4 acc[3][2][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int r = 0; r < 2; ++r) { acc[s][r][0] = sf32x4{0, 0, 0, 0}; acc[s][r][1] = sf32x4{0, 0, 0, 0}; }

    // store offsets inside one output plane (bytes); OOB = dropped by the buffer store
    unsigned yoff[2];
#pragma unroll
    for (int row = 0; row < 2; ++row) {
        const int oy = y0 + 2 * rp + row;
        yoff[row] = (col < 8 && oy < h) ? (unsigned)((((size_t)oy * w + x0 + 16 * xh + 4 * (lane >> 4)) * p.cout + cb * 8 + col) * 4) : OOB;
    }
    const int oxb = x0 + 16 * xh + 4 * (lane >> 4);
    const size_t yplane_f = (size_t)h * w * p.cout;
    const unsigned vox_b = (unsigned)p.cout * 4u;  // bytes from one output voxel to the next

    fetch(dz0 - 1);
    asm volatile("" : "+v"(esc_), "+v"(esh_), "+v"(eun_));  // the per-lane epilogue constants have landed: no vmcnt wait inside the loop
    stash(dz0 - 1);
    fetch(dz0);

    // One step = one input plane z (outputs dz0 .. dz1-1 need z = dz0-1 .. dz1), ONE stream of 108 MFMAs in two passes over the
    // plane's 24 fragments (fragment = input row rr, kw, term; rows 1 and 2, which feed both output rows, first):
    //   pass A  tap kd = 2 only: finishes output plane z-1 (36 MFMAs)
    //   pass B  taps kd = 1, 0 for planes z and z+1 (72 MFMAs); everything else rides between these MFMAs of the same wave,
    //           where about two vector instructions per MFMA issue for free (tools/micro/coissue3.hip):
    //           groups 0..7   epilogue + store of plane z-1 (one of the lane's 8 values per group)
    //           groups 8..23  split item k = (g-8)/4 of plane z+1 (loaded a step ago), a pair of floats per group, write it to
    //                         the other LDS slot, reload the item's registers with plane z+2
    // so there is no staging phase, no copy of the finished accumulators and one barrier per step.  In a phase-separated
    // version (stage, barrier, MFMAs, stores) the two workgroups of a CU ran in lockstep and the matrix pipe idled through
    // every staging phase (50 % busy).
    // With loads AND stores outstanding the compiler has to treat vmcnt as unordered: the first use of a prefetched register
    // (pass B, group 8) becomes a full flush.  By then the youngest load is two thirds of a step old and the stores a full one.
    [[maybe_unused]] float amax_ = 0.f;  // max |y| of this lane's stored values (YAMAX)
    for (int z = dz0 - 1; z <= dz1; ++z) {
        if (!(SPLIT_KO & 8)) __syncthreads();  // plane z is staged; the other slot (plane z-1) has been read by every wave
        const char* slot = ring + (z & 1) * S_PLANE_BYTES;
        char* wslot = ring + ((z + 1) & 1) * S_PLANE_BYTES;
        const int din = z + 2, dout = z - 1;
        const bool in_ok = din >= 0 && din < D && din <= dz1, out_ok = dout >= dz0 && dout < dz1;  // block-uniform
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(xb + (in_ok ? (size_t)din * plane_f : 0)), 0, in_ok ? (int)(plane_f * 4) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
            p.y + ((size_t)b * D + (out_ok ? dout : 0)) * yplane_f, 0, out_ok ? (int)(yplane_f * 4) : 0, 0x00020000);
        int ox = oxb;
        asm volatile("" : "+v"(ox));  // opaque: keeps the eight store offsets from being hoisted into eight live VGPRs

        // read i = pass * 24 + fragment; reads pipelined DEPTH deep by hand
        constexpr int NF = G::NF, NR = 2 * NF, DEPTH = SPLIT_DEPTH;
        h16x8 fr[DEPTH];
        auto rr_of = [](int ro) { return ro == 0 ? 1 : ro == 1 ? 2 : ro == 2 ? 0 : 3; };
        auto frag = [&](int i) {
            const int g = i % NF, term = g & 1, kw = (g >> 1) % KWG, rr = rr_of(g / (2 * KWG));
            if (SPLIT_KO & 32) return wf[i % NT];
            return *reinterpret_cast<const h16x8*>(slot + term * S_HALF_BYTES + fragk[kw] + rr * S_COLS * G::VB);
        };
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) fr[i] = frag(i);
        // ---- pass A
#pragma unroll
        for (int g = 0; g < NF; ++g) {
            const int term = g & 1, kw = (g >> 1) % KWG, rr = rr_of(g / (2 * KWG));
            const h16x8 f = fr[g % DEPTH];
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                const int kh = rr - row;
                if (kh < 0 || kh > 2) continue;
                if (SPLIT_KO & 16) acc[0][row][term][0] += f[0] * wf[(6 + kh) * KWG + kw][0];
                else acc[0][row][term] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f, wf[(6 + kh) * KWG + kw], acc[0][row][term], 0, 0, 0);
            }
            fr[g % DEPTH] = frag(g + DEPTH);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- pass B.  lane l holds voxels 4*(l/16) .. +3 of column l%16: columns 0..7 = hi*hi for cout c, columns 8..15 = hi*lo;
        // result(c) = acc_hi[c] + 2^-11 (acc_hi[c + 8] + acc_lo[c])
        float hl = __shfl_down(acc[0][0][0][0], 8, 16);
        unsigned hq0 = 0, lq0 = 0;
#pragma unroll
        for (int g = 0; g < NF; ++g) {
            const int term = g & 1, kw = (g >> 1) % KWG, rr = rr_of(g / (2 * KWG));
            const h16x8 f = fr[(NF + g) % DEPTH];
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                const int kh = rr - row;
                if (kh < 0 || kh > 2) continue;
#pragma unroll
                for (int s = 1; s < 3; ++s)  // plane slot s takes tap kd = 2 - s
                    if (SPLIT_KO & 16) acc[s][row][term][0] += f[0] * wf[((2 - s) * 3 + kh) * KWG + kw][0];
                    else acc[s][row][term] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f, wf[((2 - s) * 3 + kh) * KWG + kw], acc[s][row][term], 0, 0, 0);
            }
            if (NF + g + DEPTH < NR) fr[(NF + g) % DEPTH] = frag(NF + g + DEPTH);
            if (g < 8) {  // epilogue value g of the finished plane
                const int row = g >> 2, i = g & 3;
                const float cur = hl;
                if (g + 1 < 8) hl = __shfl_down(acc[0][(g + 1) >> 2][0][(g + 1) & 3], 8, 16);
                float r = __builtin_fmaf(cur + acc[0][row][1][i], 1.0f / 2048.0f, acc[0][row][0][i]);
                r = fmaxf(__builtin_fmaf(r * eun_, esc_, esh_), floor_);
                if constexpr (YAMAX)
                    if (out_ok && yoff[row] != OOB && ox + i < w) amax_ = fmaxf(amax_, finite_abs_or_zero(r));
                if (!(SPLIT_KO & 2) || r == 12345.678f)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), rs_out, ox + i < w ? yoff[row] + vox_b * i : OOB, 0, 0);
            } else if (!(SPLIT_KO & 4)) {  // staging of plane z+1 / reload with plane z+2: item k, float pair q of its 4
                const int k = (g - 8) >> 2, q = (g - 8) & 3;
                const sf32x4 v = __builtin_bit_cast(sf32x4, pre[k][q >> 1]);
                unsigned hq1, lq1;
                split2(v[(q & 1) * 2], v[(q & 1) * 2 + 1], hq1, lq1);
                if (q & 1) {
                    char* dst = (loff[k] >= 0 ? wslot + loff[k] : dump) + 8 * (q >> 1);
                    *reinterpret_cast<su32x2*>(dst) = su32x2{hq0, hq1};
                    *reinterpret_cast<su32x2*>(loff[k] >= 0 ? dst + S_HALF_BYTES : dst) = su32x2{lq0, lq1};
                } else {
                    hq0 = hq1; lq0 = lq1;
                }
                if (q == 3 && !(SPLIT_KO & 1)) {
                    pre[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, gob[k], 0, 0);
                    pre[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, gob[k] + 16u, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                acc[0][r][t] = acc[1][r][t];
                acc[1][r][t] = acc[2][r][t];
                acc[2][r][t] = sf32x4{0, 0, 0, 0};
            }
    }
    if constexpr (YAMAX) {  // what a following split-operand layer scales its activations by: ONE atomic per workgroup
        for (int o = 32; o > 0; o >>= 1) amax_ = fmaxf(amax_, __shfl_xor(amax_, o));
        __syncthreads();  // the ring is free
        float* wm = reinterpret_cast<float*>(ring);
        if (lane == 0) wm[wv] = amax_;
        __syncthreads();
        if (tid == 0) {
            amax_ = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
            raise_absmax(p.yamax, amax_);
        }
    }
}

}  // namespace mvd

static bool split_shape_ok(int Cin, int Cout) { return (Cin == 32 || Cin == 16) && Cout >= 8 && Cout <= 64 && Cout % 8 == 0; }
static size_t split_frag_bytes(int Cin, int Cout) { return (size_t)(Cout / 8) * (Cin == 32 ? 27 : 18) * 64 * 16; }

template <int CIN, bool YAMAX>
static int launch_split_v(const mvd::SplitParams& p, long long nblk, hipStream_t st) {
    const size_t lds = 2 * (size_t)mvd::SplitGeom<CIN>::PLANE_BYTES + 256 * 16;  // two slots + the dump row
    (void)hipFuncSetAttribute((const void*)mvd::conv0_split_kernel<CIN, YAMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((mvd::conv0_split_kernel<CIN, YAMAX>), dim3((unsigned)nblk), dim3(256), lds, st, p);
    return mvd::launch_status("conv3d_split");
}
template <int CIN>
static int launch_split(const mvd::SplitParams& p, long long nblk, hipStream_t st) {
    return p.yamax ? launch_split_v<CIN, true>(p, nblk, st) : launch_split_v<CIN, false>(p, nblk, st);
}

extern "C" {

size_t mvd_conv3d_split_packed_weight_bytes(int Cin, int Cout) {
    return split_shape_ok(Cin, Cout) ? split_frag_bytes(Cin, Cout) + (size_t)Cout * sizeof(float) : 0;
}

int mvd_pack_conv3d_weights_split(const float* w, int Cin, int Cout, void* packed, mvd_stream_t stream) {
    MVD_REQUIRE(w && packed, "pack_conv3d_weights_split: NULL argument");
    MVD_REQUIRE(split_shape_ok(Cin, Cout), "pack_conv3d_weights_split: 16 or 32 input channels and 8, 16, ... 64 output channels are built (got %d -> %d)", Cin, Cout);
    const int ncb = Cout / 8;
    float* wscale = reinterpret_cast<float*>(static_cast<char*>(packed) + split_frag_bytes(Cin, Cout));
    hipLaunchKernelGGL(mvd::conv0_split_wscale_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, w, wscale, Cin);
    const unsigned nb = (unsigned)((split_frag_bytes(Cin, Cout) / 2 + 255) / 256);
    if (Cin == 32) hipLaunchKernelGGL(mvd::pack_conv0_split_kernel<32>, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, wscale, (_Float16*)packed, ncb);
    else hipLaunchKernelGGL(mvd::pack_conv0_split_kernel<16>, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, wscale, (_Float16*)packed, ncb);
    return mvd::launch_status("pack_conv3d_weights_split");
}

static int conv3d_split_entry(const float* x, const float* x_absmax, const void* packed_w, const float* scale, const float* shift, float* y,
                              float* y_absmax, int B, int D, int h, int w, int Cin, int Cout, int relu, mvd_stream_t stream);

int mvd_conv3d_bn_relu_f32_split(const float* x, const float* x_absmax, const void* packed_w, const float* scale, const float* shift,
                                 float* y, int B, int D, int h, int w, int Cin, int Cout, int relu, mvd_stream_t stream) {
    return conv3d_split_entry(x, x_absmax, packed_w, scale, shift, y, nullptr, B, D, h, w, Cin, Cout, relu, stream);
}

int mvd_conv3d_bn_relu_absmax_f32_split(const float* x, const float* x_absmax, const void* packed_w, const float* scale, const float* shift,
                                        float* y, float* y_absmax, int B, int D, int h, int w, int Cin, int Cout, int relu,
                                        mvd_stream_t stream) {
    MVD_REQUIRE(y_absmax, "conv3d_split_absmax: NULL argument");
    return conv3d_split_entry(x, x_absmax, packed_w, scale, shift, y, y_absmax, B, D, h, w, Cin, Cout, relu, stream);
}

static int conv3d_split_entry(const float* x, const float* x_absmax, const void* packed_w, const float* scale, const float* shift, float* y,
                              float* y_absmax, int B, int D, int h, int w, int Cin, int Cout, int relu, mvd_stream_t stream) {
    MVD_REQUIRE(x && x_absmax && packed_w && scale && shift && y, "conv3d_split: NULL argument");
    MVD_REQUIRE(split_shape_ok(Cin, Cout), "conv3d_split: 16 or 32 input channels and 8, 16, ... 64 output channels are built (got %d -> %d)", Cin, Cout);
    MVD_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "conv3d_split: non-positive dimension");
    MVD_REQUIRE((long long)h * w * Cin * 4 < 0x7fffffffLL && (long long)h * w * Cout * 4 < 0x7fffffffLL,
                "conv3d_split: one plane exceeds the 31-bit byte-offset range");
    mvd::SplitParams p{};
    p.x = x; p.absmax = x_absmax; p.wpk = (const char*)packed_w; p.scale = scale; p.shift = shift; p.y = y; p.yamax = y_absmax;
    if (y_absmax && hipMemsetAsync(y_absmax, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) return mvd::launch_status("conv3d_split: memset");
    p.B = B; p.D = D; p.h = h; p.w = w; p.relu = relu;
    p.cout = Cout; p.ncb = Cout / 8;
    p.tiles_x = (w + mvd::S_TW - 1) / mvd::S_TW;
    p.tiles_y = (h + mvd::S_TH - 1) / mvd::S_TH;
    const long long tiles = (long long)p.tiles_x * p.tiles_y;
    p.tiles_per_xcd = (int)((tiles + 7) / 8);
    int td = 32;
    while (td > 8 && tiles * B * p.ncb * ((D + td - 1) / td) < 2048) td /= 2;
    p.td = td;
    p.dgroups = (D + td - 1) / td;
    const long long nblk = 8LL * p.ncb * p.tiles_per_xcd * p.dgroups * B;
    MVD_REQUIRE(nblk <= 0x7fffffffLL, "conv3d_split: %lld workgroups exceed the grid limit", nblk);
    return Cin == 32 ? launch_split<32>(p, nblk, (hipStream_t)stream) : launch_split<16>(p, nblk, (hipStream_t)stream);
}
}
