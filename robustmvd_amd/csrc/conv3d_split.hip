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
// Workgroup (4 waves) = 4 x 32 tile marching through TD + 2 input planes (see the kernel); the fp32 volume is converted to
// the two fp16 terms while it is staged (global -> registers -> LDS), all 27 weight fragments stay in registers.
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

// Plane-stationary march.  One input plane z is staged per step and every activation fragment read from LDS is used for all
// the outputs it feeds in d and y: the three output planes z-1, z, z+1 (taps kd = 2, 1, 0) and the wave's two output rows
// (taps kh = rr - row).  24 fragment reads feed the 108 MFMAs of a step (4.5 per read); with one read per MFMA (the first
// version, profiles/r02_conv0_split_pmc.txt) the four SIMDs of a CU asked the LDS for 32 clocks of ds_read_b128 per 16
// clocks of MFMA and the kernel sat on the LDS port at 35 % matrix utilisation.  Three output planes of accumulators are in
// flight per wave (48 VGPRs); the LDS holds two plane slots (the one being read, the one being written).
// wave wv: columns 16 (wv & 1) .. +15, output rows 2 (wv >> 1), +1 of the 4 x 32 tile.
__global__ void __launch_bounds__(256, 2) conv0_split_kernel(SplitParams p) {
    extern __shared__ __attribute__((aligned(16))) char ring[];  // 2 planes x (hi part | lo part)
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
        const int vox = e >> 2, ch = e & 3;
        const int r = vox / S_COLS, c = vox - r * S_COLS;
        const int gy = y0 - 1 + r, gx = x0 - 1 + c;
        const bool in = e < S_ITEMS && gy >= 0 && gy < h && gx >= 0 && gx < w;
        gob[k] = in ? (unsigned)(((gy * w + gx) * 32 + ch * 8) * 4) : OOB;                  // bytes inside a plane
        loff[k] = e < S_ITEMS ? (vox * 64 + ((ch ^ ((c >> 1) & 3)) * 16)) : -1;             // swizzled as in conv0_f16
    }
    const size_t plane_f = (size_t)h * w * 32;
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
    auto stash = [&](int d) {  // split into the two fp16 terms on the way into the LDS
        char* slot = ring + (d & 1) * S_PLANE_BYTES;
#pragma unroll
        for (int k = 0; k < S_NLOAD; ++k) {
            const sf32x4 a = __builtin_bit_cast(sf32x4, pre[k][0]), c = __builtin_bit_cast(sf32x4, pre[k][1]);
            const _Float16 h0 = (_Float16)a.x, h1 = (_Float16)a.y, h2 = (_Float16)a.z, h3 = (_Float16)a.w;
            const _Float16 h4 = (_Float16)c.x, h5 = (_Float16)c.y, h6 = (_Float16)c.z, h7 = (_Float16)c.w;
            const su32x4 hv = {pack_h2(h0, h1), pack_h2(h2, h3), pack_h2(h4, h5), pack_h2(h6, h7)};
            const su32x4 lv = {pack_h2((_Float16)((a.x - (float)h0) * 2048.0f), (_Float16)((a.y - (float)h1) * 2048.0f)),
                               pack_h2((_Float16)((a.z - (float)h2) * 2048.0f), (_Float16)((a.w - (float)h3) * 2048.0f)),
                               pack_h2((_Float16)((c.x - (float)h4) * 2048.0f), (_Float16)((c.y - (float)h5) * 2048.0f)),
                               pack_h2((_Float16)((c.z - (float)h6) * 2048.0f), (_Float16)((c.w - (float)h7) * 2048.0f))};
            char* dst = loff[k] >= 0 ? slot + loff[k] : dump;
            *reinterpret_cast<su32x4*>(dst) = hv;
            *reinterpret_cast<su32x4*>(loff[k] >= 0 ? dst + S_HALF_BYTES : dst) = lv;
        }
    };

    // epilogue constants of the lanes that end up with a result: column (cout) l%16 < 8
    const int col = lane & 15;
    float esc_ = p.scale[col & 7], esh_ = p.shift[col & 7];
    const int xh = wv & 1, rp = wv >> 1;

    int fragk[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int c = 16 * xh + (lane & 15) + kw;
        fragk[kw] = (2 * rp * S_COLS + c) * 64 + (((lane >> 4) ^ ((c >> 1) & 3)) * 16);
    }

    // acc[plane slot][row][term]: slot 0 = output plane z-1 (finishes this step), 1 = plane z, 2 = plane z+1 (starts);
    // fin = the plane that finished in the previous step, stored after this step's barrier
    sf32x4 acc[3][2][2], fin[2][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int r = 0; r < 2; ++r) { acc[s][r][0] = sf32x4{0, 0, 0, 0}; acc[s][r][1] = sf32x4{0, 0, 0, 0}; }

    // lane l holds voxels 4*(l/16) .. +3 of column l%16: columns 0..7 = hi*hi for cout c, columns 8..15 = hi*lo;
    // result(c) = acc_hi[c] + 2^-11 (acc_hi[c + 8] + acc_lo[c]).  Called for the finished plane AFTER the next step's barrier
    // and BEFORE its prefetch: vmcnt counts loads and stores in order, so stores issued after the prefetch would sit between
    // the loads and the wait that the next staging needs (the compiler then waits for the store round trip, vmcnt(0)).
    // store offsets inside one output plane (bytes); OOB = dropped by the buffer store
    unsigned yoff[2];
#pragma unroll
    for (int row = 0; row < 2; ++row) {
        const int oy = y0 + 2 * rp + row;
        yoff[row] = (col < 8 && oy < h) ? (unsigned)((((size_t)oy * w + x0 + 16 * xh + 4 * (lane >> 4)) * 8 + col) * 4) : OOB;
    }
    const int oxb = x0 + 16 * xh + 4 * (lane >> 4);
    const size_t yplane_f = (size_t)h * w * 8;
    auto emit = [&](int d, const sf32x4 (&a)[2][2]) {
        const bool din = d >= dz0 && d < dz1;  // block-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            p.y + ((size_t)b * D + (din ? d : 0)) * yplane_f, 0, din ? (int)(yplane_f * 4) : 0, 0x00020000);
        int ox = oxb;
        asm volatile("" : "+v"(ox));  // opaque: keeps the eight store offsets from being hoisted into eight live VGPRs
#pragma unroll
        for (int row = 0; row < 2; ++row) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float hl = __shfl_down(a[row][0][i], 8, 16);
                float r = a[row][0][i] + (hl + a[row][1][i]) * (1.0f / 2048.0f);
                r = fmaf(r, esc_, esh_);
                if (p.relu) r = fmaxf(r, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), rs, ox + i < w ? yoff[row] + 32u * i : OOB, 0, 0);
            }
        }
    };

    fetch(dz0 - 1);
    asm volatile("" : "+v"(esc_), "+v"(esh_));  // the per-lane epilogue constants have landed: no vmcnt wait inside the loop
    for (int z = dz0 - 1; z <= dz1; ++z) {  // input planes; outputs dz0 .. dz1-1
        stash(z);
        __syncthreads();  // also orders this step's writes of slot z&1 after the reads of step z-2
        emit(z - 2, fin);
        fetch(z + 1);
        const char* slot = ring + (z & 1) * S_PLANE_BYTES;
        // fragment i = (input row rr, kw, term): 24 per plane, reads pipelined DEPTH deep by hand
        constexpr int NF = 24, DEPTH = 4;
        h16x8 fr[DEPTH];
        auto frag = [&](int i) {
            const int term = i & 1, kw = (i >> 1) % 3, rr = i / 6;
            return *reinterpret_cast<const h16x8*>(slot + term * S_HALF_BYTES + fragk[kw] + rr * S_COLS * 64);
        };
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) fr[i] = frag(i);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
            for (int kt = 0; kt < 6; ++kt) {
                const int i = rr * 6 + kt, term = kt & 1, kw = kt >> 1;
                const h16x8 f = fr[i % DEPTH];
#pragma unroll
                for (int row = 0; row < 2; ++row) {
                    const int kh = rr - row;
                    if (kh < 0 || kh > 2) continue;
#pragma unroll
                    for (int s = 0; s < 3; ++s)  // plane slot s takes tap kd = 2 - s
                        acc[s][row][term] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f, wf[(2 - s) * 9 + kh * 3 + kw], acc[s][row][term], 0, 0, 0);
                }
                if (i + DEPTH < NF) fr[i % DEPTH] = frag(i + DEPTH);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fin[r][t] = acc[0][r][t];
                acc[0][r][t] = acc[1][r][t];
                acc[1][r][t] = acc[2][r][t];
                acc[2][r][t] = sf32x4{0, 0, 0, 0};
            }
    }
    emit(dz1 - 1, fin);
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
    MVD_REQUIRE((long long)h * w * 128 < 0x7fffffffLL, "conv3d_split: one input plane exceeds the 31-bit byte-offset range");
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
    const size_t lds = 2 * (size_t)mvd::S_PLANE_BYTES + 256 * 16;  // two slots + the dump row
    (void)hipFuncSetAttribute((const void*)mvd::conv0_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mvd::conv0_split_kernel, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, p);
    return mvd::launch_status("conv3d_split");
}
}
