// K4 first layer with fp32 operands split into two fp16 terms each (OPT-IN; the default conv0 stays on fp32 MFMA).
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
// Workgroup (4 waves) = 4 x 32 tile marching through TD planes with a 3-plane LDS ring; the fp32 volume is converted to the
// two fp16 terms while it is staged (global -> registers -> LDS), all 27 weight fragments stay in registers.
#include "mvd_common.h"

namespace mvd {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float sf32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int su32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

constexpr int S_TH = 4, S_TW = 32, S_ROWS = S_TH + 2, S_COLS = S_TW + 2;
constexpr int S_HALF_BYTES = S_ROWS * S_COLS * 64;      // hi (or lo) part of one staged plane
constexpr int S_PLANE_BYTES = 2 * S_HALF_BYTES;
constexpr int S_ITEMS = S_ROWS * S_COLS * 4;             // (voxel, 8-channel chunk) items per plane
constexpr int S_NLOAD = (S_ITEMS + 255) / 256;

// w (8, 32, 3, 3, 3) fp32 -> [tap 27][lane 64][8 halves]: lane l = column l%16 (0..7: w_hi of cout l%16, 8..15: w_lo of cout
// l%16 - 8), cin 8*(l/16) .. +7
__global__ void pack_conv0_split_kernel(const float* __restrict__ w, _Float16* __restrict__ packed) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 27 * 64 * 8) return;
    const int j = e & 7, lane = (e >> 3) & 63, tap = e >> 9;
    const int col = lane & 15, cin = 8 * (lane >> 4) + j, cout = col & 7;
    const float v = w[((size_t)cout * 32 + cin) * 27 + tap];
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)((v - (float)hi) * 2048.0f);
    packed[e] = col < 8 ? hi : lo;
}

struct SplitParams {
    const float* x;       // (B, D, h, w, 32) fp32
    const char* wpk;
    const float* scale;
    const float* shift;
    float* y;             // (B, D, h, w, 8) fp32
    int B, D, h, w, relu;
    int tiles_x, tiles_y, dgroups, td, tiles_per_xcd;
};

__device__ __forceinline__ unsigned pack_h2(_Float16 a, _Float16 b) {
    const h16x2 v = {a, b};
    return __builtin_bit_cast(unsigned, v);
}

__global__ void __launch_bounds__(256, 2) conv0_split_kernel(SplitParams p) {
    extern __shared__ __attribute__((aligned(16))) char ring[];  // 3 planes x (hi part | lo part)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int D = p.D, h = p.h, w = p.w;

    // XCD-aware decode (blocks b and b + 8 share an XCD and its L2): each XCD owns a contiguous run of (row-major) tiles, so
    // the halo rows and columns that neighbouring tiles share are re-read from ONE L2 instead of from HBM by eight
    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    const int t_in = j % p.tiles_per_xcd; j /= p.tiles_per_xcd;
    const int dg = j % p.dgroups;
    const int b = j / p.dgroups;
    const int tile = xcd * p.tiles_per_xcd + t_in;
    if (tile >= p.tiles_x * p.tiles_y) return;  // block-uniform
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const int x0 = tx * S_TW, y0 = ty * S_TH;
    const int dz0 = dg * p.td, dz1 = min(dz0 + p.td, D);

    h16x8 wf[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) wf[t] = *reinterpret_cast<const h16x8*>(p.wpk + ((size_t)t * 64 + lane) * 16);

    // staging map: item e = tid + 256 k -> (row, col, 8-channel chunk)
    int goff[S_NLOAD], loff[S_NLOAD];
#pragma unroll
    for (int k = 0; k < S_NLOAD; ++k) {
        const int e = tid + 256 * k;
        const int vox = e >> 2, ch = e & 3;
        const int r = vox / S_COLS, c = vox - r * S_COLS;
        const int gy = y0 - 1 + r, gx = x0 - 1 + c;
        const bool in = e < S_ITEMS && gy >= 0 && gy < h && gx >= 0 && gx < w;
        goff[k] = in ? ((gy * w + gx) * 32 + ch * 8) : -1;                                   // floats inside a plane
        loff[k] = e < S_ITEMS ? (vox * 64 + ((ch ^ ((c >> 1) & 3)) * 16)) : -1;             // swizzled as in conv0_f16
    }
    const size_t plane_f = (size_t)h * w * 32;
    const float* xb = p.x + (size_t)b * D * plane_f;
    float4 pre[S_NLOAD][2];
    auto fetch = [&](int d) {
        const bool din = d >= 0 && d < D;  // block-uniform
#pragma unroll
        for (int k = 0; k < S_NLOAD; ++k) {
            pre[k][0] = pre[k][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (din && goff[k] >= 0) {
                const float4* src = reinterpret_cast<const float4*>(xb + (size_t)d * plane_f + goff[k]);
                pre[k][0] = src[0];
                pre[k][1] = src[1];
            }
        }
    };
    auto stash = [&](int d) {  // split into the two fp16 terms on the way into the ring
        char* slot = ring + ((d + 3) % 3) * S_PLANE_BYTES;
#pragma unroll
        for (int k = 0; k < S_NLOAD; ++k) {
            const float4 a = pre[k][0], c = pre[k][1];
            const _Float16 h0 = (_Float16)a.x, h1 = (_Float16)a.y, h2 = (_Float16)a.z, h3 = (_Float16)a.w;
            const _Float16 h4 = (_Float16)c.x, h5 = (_Float16)c.y, h6 = (_Float16)c.z, h7 = (_Float16)c.w;
            const su32x4 hv = {pack_h2(h0, h1), pack_h2(h2, h3), pack_h2(h4, h5), pack_h2(h6, h7)};
            const su32x4 lv = {pack_h2((_Float16)((a.x - (float)h0) * 2048.0f), (_Float16)((a.y - (float)h1) * 2048.0f)),
                               pack_h2((_Float16)((a.z - (float)h2) * 2048.0f), (_Float16)((a.w - (float)h3) * 2048.0f)),
                               pack_h2((_Float16)((c.x - (float)h4) * 2048.0f), (_Float16)((c.y - (float)h5) * 2048.0f)),
                               pack_h2((_Float16)((c.z - (float)h6) * 2048.0f), (_Float16)((c.w - (float)h7) * 2048.0f))};
            if (loff[k] >= 0) {
                *reinterpret_cast<su32x4*>(slot + loff[k]) = hv;
                *reinterpret_cast<su32x4*>(slot + S_HALF_BYTES + loff[k]) = lv;
            }
        }
    };

    // epilogue constants of the lanes that end up with a result: column (cout) l%16 < 8
    const int col = lane & 15;
    const float esc = p.scale[col & 7], esh = p.shift[col & 7];
    const int oy = y0 + wv;

    fetch(dz0 - 1); stash(dz0 - 1);
    fetch(dz0);     stash(dz0);
    fetch(dz0 + 1);
    int fragk[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int c = (lane & 15) + kw;
        fragk[kw] = (wv * S_COLS + c) * 64 + (((lane >> 4) ^ ((c >> 1) & 3)) * 16);
    }

    for (int d = dz0; d < dz1; ++d) {
        stash(d + 1);
        __syncthreads();
        fetch(d + 2);
        sf32x4 acc1[2], acc2[2];  // a_hi x [w_hi | w_lo],  a_lo x [w_hi | (w_lo: unused)]
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) { acc1[cg] = sf32x4{0, 0, 0, 0}; acc2[cg] = sf32x4{0, 0, 0, 0}; }
        const char* slots[3] = {ring + ((d + 2) % 3) * S_PLANE_BYTES, ring + (d % 3) * S_PLANE_BYTES,
                                ring + ((d + 1) % 3) * S_PLANE_BYTES};
        // product i = (tap, column group, term): 27 x 2 x 2 = 108 per plane, fragment reads pipelined 8 deep by hand
        constexpr int NP = 108, DEPTH = 8;
        h16x8 fr[DEPTH];
        auto frag = [&](int i) {
            const int term = i & 1, cg = (i >> 1) & 1, tap = i >> 2, kw = tap % 3, kh = (tap / 3) % 3, kd = tap / 9;
            return *reinterpret_cast<const h16x8*>(slots[kd] + term * S_HALF_BYTES + fragk[kw] + (kh * S_COLS + cg * 16) * 64);
        };
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) fr[i] = frag(i);
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int i = tap * 4 + s4, term = s4 & 1, cg = s4 >> 1;
                if (term == 0) acc1[cg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[i % DEPTH], wf[tap], acc1[cg], 0, 0, 0);
                else acc2[cg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[i % DEPTH], wf[tap], acc2[cg], 0, 0, 0);
                if (i + DEPTH < NP) fr[i % DEPTH] = frag(i + DEPTH);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // lane l holds rows (voxels) 4*(l/16) .. +3 of column l%16: columns 0..7 = hi*hi for cout c, columns 8..15 = hi*lo;
        // result(c) = acc1[c] + 2^-11 (acc1[c + 8] + acc2[c])
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
            float r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float hl = __shfl_down(acc1[cg][i], 8, 16);
                r[i] = acc1[cg][i] + (hl + acc2[cg][i]) * (1.0f / 2048.0f);
                r[i] = fmaf(r[i], esc, esh);
                if (p.relu) r[i] = fmaxf(r[i], 0.f);
            }
            if (col < 8 && oy < h) {
                float* yrow = p.y + ((((size_t)b * D + d) * h + oy) * w) * 8 + col;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ox = x0 + cg * 16 + 4 * (lane >> 4) + i;
                    if (ox < w) yrow[(size_t)ox * 8] = r[i];
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace mvd

extern "C" {

size_t mvd_conv3d_split_packed_weight_bytes(int Cin, int Cout) { return (Cin == 32 && Cout == 8) ? (size_t)27 * 64 * 16 : 0; }

int mvd_pack_conv3d_weights_split(const float* w, int Cin, int Cout, void* packed, mvd_stream_t stream) {
    MVD_REQUIRE(w && packed, "pack_conv3d_weights_split: NULL argument");
    MVD_REQUIRE(Cin == 32 && Cout == 8, "pack_conv3d_weights_split: only the 32 -> 8 first layer of CostRegNet is built (got %d -> %d)", Cin, Cout);
    hipLaunchKernelGGL(mvd::pack_conv0_split_kernel, dim3((27 * 64 * 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                       (_Float16*)packed);
    return mvd::launch_status("pack_conv3d_weights_split");
}

int mvd_conv3d_bn_relu_f32_split(const float* x, const void* packed_w, const float* scale, const float* shift, float* y, int B,
                                 int D, int h, int w, int Cin, int Cout, int relu, mvd_stream_t stream) {
    MVD_REQUIRE(x && packed_w && scale && shift && y, "conv3d_split: NULL argument");
    MVD_REQUIRE(Cin == 32 && Cout == 8, "conv3d_split: only the 32 -> 8 first layer of CostRegNet is built (got %d -> %d)", Cin, Cout);
    MVD_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "conv3d_split: non-positive dimension");
    MVD_REQUIRE((long long)h * w * 32 < 0x7fffffffLL, "conv3d_split: one input plane exceeds the 32-bit offset range");
    mvd::SplitParams p{};
    p.x = x; p.wpk = (const char*)packed_w; p.scale = scale; p.shift = shift; p.y = y;
    p.B = B; p.D = D; p.h = h; p.w = w; p.relu = relu;
    p.tiles_x = (w + mvd::S_TW - 1) / mvd::S_TW;
    p.tiles_y = (h + mvd::S_TH - 1) / mvd::S_TH;
    const long long tiles = (long long)p.tiles_x * p.tiles_y;
    p.tiles_per_xcd = (int)((tiles + 7) / 8);
    int td = 32;
    while (td > 8 && tiles * B * ((D + td - 1) / td) < 2048) td /= 2;
    p.td = td;
    p.dgroups = (D + td - 1) / td;
    const long long nblk = 8LL * p.tiles_per_xcd * p.dgroups * B;
    MVD_REQUIRE(nblk <= 0x7fffffffLL, "conv3d_split: %lld workgroups exceed the grid limit", nblk);
    const size_t lds = 3 * (size_t)mvd::S_PLANE_BYTES;
    (void)hipFuncSetAttribute((const void*)mvd::conv0_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mvd::conv0_split_kernel, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, p);
    return mvd::launch_status("conv3d_split");
}
}
