// K4, fp16-input first layer (BASELINE.json configs[3]: "3D-conv regulariser on MFMA, fp16 features").
// conv0 of CostRegNet (rmvd/models/blocks/mvsnet_components.py:78, ConvBnReLU3D 32 -> 8, 3x3x3, stride 1, padding 1;
// :25-41) reading the fp16 variance volume that mvd_warp_variance_f16 writes and producing the fp32 (B,D,h,w,8)
// activations the rest of the (fp32-MFMA) regulariser consumes.  68 % of the regulariser's FLOPs sit in this layer; on
// v_mfma_f32_16x16x32_f16 (fp16 operands, fp32 accumulate) it needs 1/16 of the matrix-pipe cycles of the fp32 form and
// becomes a memory-bound pass: 2 B/channel in, 4 B/channel out.
//
// Implicit GEMM per kernel tap: D[cout, voxel] += W_tap[cout, cin] * X[cin, voxel + tap]; A = weights (16 rows, couts
// 0..7 valid), B = activations of 16 consecutive voxels of one row, K = the 32 input channels in ONE instruction.
// The C/D layout gives lane l the 4 consecutive couts 4*(l/16).. of voxel l%16: lanes 0..31 store one float4 each and a
// wave's store covers 16 voxels x 32 B contiguously.
//
// A workgroup (4 waves) owns a 4 x 64 tile of (y, x) and marches through TD planes with a 3-plane ring of the input
// tile (+ halo, zero outside the volume = the conv's padding) in LDS; the next plane is fetched into registers under
// the MFMAs.  All 27 weight fragments stay in registers (108 of the 256 a wave has at two waves per SIMD).
#include "mvd_common.h"

namespace mvd {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

constexpr int C0_TH = 4, C0_TW = 64, C0_ROWS = C0_TH + 2, C0_COLS = C0_TW + 2;
constexpr int C0_PLANE_CHUNKS = C0_ROWS * C0_COLS * 4;            // 16-byte chunks per staged plane
constexpr int C0_NLOAD = (C0_PLANE_CHUNKS + 255) / 256;           // chunks per thread
constexpr int C0_PLANE_BYTES = C0_ROWS * C0_COLS * 64;

// w (8, 32, 3, 3, 3) fp32 -> fragment order [tap 27][lane 64][8 halves]: lane l = (cout l%16, cin 8*(l/16) .. +7)
__global__ void pack_conv0_f16_kernel(const float* __restrict__ w, _Float16* __restrict__ packed) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 27 * 64 * 8) return;
    const int j = e & 7, lane = (e >> 3) & 63, tap = e >> 9;
    const int cout = lane & 15, cin = 8 * (lane >> 4) + j;
    packed[e] = cout < 8 ? (_Float16)w[((size_t)cout * 32 + cin) * 27 + tap] : (_Float16)0.0f;
}

struct Conv0Params {
    const char* x;        // (B, D, h, w, 32) fp16
    const char* wpk;      // packed weights
    const float* scale;   // (8)
    const float* shift;   // (8)
    float* y;             // (B, D, h, w, 8) fp32
    int B, D, h, w, relu;
    int tiles_x, tiles_y, dgroups, td, tiles_per_xcd;
};

__global__ void __launch_bounds__(256, 2) conv0_f16_kernel(Conv0Params p) {
    extern __shared__ __attribute__((aligned(16))) char ring[];  // 3 planes x C0_PLANE_BYTES
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int D = p.D, h = p.h, w = p.w;

    // block -> (batch, depth group, tile); tiles fastest so that neighbouring tiles (shared halo rows) run together
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
    const int x0 = tx * C0_TW, y0 = ty * C0_TH;
    const int dz0 = dg * p.td, dz1 = min(dz0 + p.td, D);

    // ---- weights: 27 fragments of 8 halves per lane, resident for the whole march ----
    f16x8 wf[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) wf[t] = *reinterpret_cast<const f16x8*>(p.wpk + ((size_t)t * 64 + lane) * 16);

    // ---- staging map of this thread: chunk e = tid + 256 k -> (row, col, 16-byte chunk) of the tile + halo ----
    // Everything on the memory side is branch-free: a voxel outside the image (or a plane outside the volume) is an
    // out-of-range buffer offset (reads 0), rows / columns beyond the output are out-of-range store offsets (dropped).  Behind
    // exec-masked branches the compiler cannot count outstanding operations and waits vmcnt(0), stores included.
    constexpr unsigned OOB = 0x80000000u;
    unsigned goff[C0_NLOAD];  // byte offset inside a plane of x, or OOB
    int loff[C0_NLOAD];       // byte offset inside a ring slot
#pragma unroll
    for (int k = 0; k < C0_NLOAD; ++k) {
        const int e = tid + 256 * k;
        const int vox = e >> 2, ch = e & 3;
        const int r = vox / C0_COLS, c = vox - r * C0_COLS;
        const int gy = y0 - 1 + r, gx = x0 - 1 + c;
        const bool in = e < C0_PLANE_CHUNKS && gy >= 0 && gy < h && gx >= 0 && gx < w;
        goff[k] = in ? (unsigned)((gy * w + gx) * 64 + ch * 16) : OOB;
        // the four 16-byte channel chunks of a voxel are XOR-swizzled by the voxel-pair index of its column: a fragment read
        // (16 consecutive voxels x 4 chunks, serviced in the four 16-lane groups of ds_read_b128) then touches every bank
        // once instead of twice (enumerated over all alignments; plain 64-B pitch: two-way conflicts everywhere)
        loff[k] = e < C0_PLANE_CHUNKS ? (vox * 64 + ((ch ^ ((c >> 1) & 3)) * 16)) : 3 * C0_PLANE_BYTES + tid * 16;  // else: dump row
    }
    const size_t plane_bytes = (size_t)h * w * 64;
    const char* xb = p.x + (size_t)b * D * plane_bytes;
    u32x4v pre[C0_NLOAD];
    auto fetch = [&](int d) {  // plane d of the input tile -> registers (zeros outside the volume)
        const bool din = d >= 0 && d < D;  // block-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(xb + (din ? (size_t)d * plane_bytes : 0)), 0, din ? (int)plane_bytes : 0, 0x00020000);
#pragma unroll
        for (int k = 0; k < C0_NLOAD; ++k) pre[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[k], 0, 0);
    };
    auto stash = [&](int d) {  // registers -> ring slot of plane d
        char* slot = ring + ((d + 3) % 3) * C0_PLANE_BYTES;
#pragma unroll
        for (int k = 0; k < C0_NLOAD; ++k)
            *reinterpret_cast<u32x4v*>(loff[k] < 3 * C0_PLANE_BYTES ? slot + loff[k] : ring + loff[k]) = pre[k];
    };

    // per-lane epilogue constants: couts 4*(lane/16) .. +3 (lanes 32..63 hold the padding rows 8..15)
    const int cq = (lane >> 4) & 1;
    const float4 sc = reinterpret_cast<const float4*>(p.scale)[cq], sh = reinterpret_cast<const float4*>(p.shift)[cq];
    const int oy = y0 + wv;
    // store offsets inside one output plane (bytes), OOB = dropped
    unsigned yoff[4];
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {
        const int ox = x0 + cg * 16 + (lane & 15);
        yoff[cg] = (oy < h && lane < 32 && ox < w) ? (unsigned)((((size_t)oy * w + ox) * 8 + cq * 4) * 4) : OOB;
    }
    const size_t yplane = (size_t)h * w * 8;
    f32x4 fin[4];  // the plane finished in the previous step: stored after this step's first barrier, BEFORE its prefetch —
                   // vmcnt retires in order, so stores issued behind the prefetch would sit between the loads and the wait
                   // the next staging needs
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) fin[cg] = f32x4{0, 0, 0, 0};
    auto emit = [&](int d) {
        const bool ok = d >= dz0 && d < dz1;  // block-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            p.y + ((size_t)b * D + (ok ? d : 0)) * yplane, 0, ok ? (int)(yplane * 4) : 0, 0x00020000);
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) {
            float r0 = fmaf(fin[cg][0], sc.x, sh.x), r1 = fmaf(fin[cg][1], sc.y, sh.y), r2 = fmaf(fin[cg][2], sc.z, sh.z),
                  r3 = fmaf(fin[cg][3], sc.w, sh.w);
            if (p.relu) { r0 = fmaxf(r0, 0.f); r1 = fmaxf(r1, 0.f); r2 = fmaxf(r2, 0.f); r3 = fmaxf(r3, 0.f); }
            __builtin_amdgcn_raw_buffer_store_b128(u32x4v{__float_as_uint(r0), __float_as_uint(r1), __float_as_uint(r2), __float_as_uint(r3)},
                                                   rs, yoff[cg], 0, 0);
        }
    };

    fetch(dz0 - 1); stash(dz0 - 1);
    fetch(dz0);     stash(dz0);
    fetch(dz0 + 1);
    // fragment address of this lane inside a slot for (row wv + kh, column group cg, kw): voxel column cg*16 + lane%16 + kw,
    // chunk lane/16 at its swizzled position (cg*16 does not change the swizzle: 16/2 = 0 mod 4)
    int fragk[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int c = (lane & 15) + kw;
        fragk[kw] = (wv * C0_COLS + c) * 64 + (((lane >> 4) ^ ((c >> 1) & 3)) * 16);
    }

    for (int d = dz0; d < dz1; ++d) {
        stash(d + 1);
        __syncthreads();          // planes d-1, d, d+1 are in the ring
        emit(d - 1);
        fetch(d + 2);             // flies under the MFMAs below
        f32x4 acc[4];
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) acc[cg] = f32x4{0, 0, 0, 0};
        // 108 (tap, column group) products per plane, column group fastest (4 independent accumulators back to back).  The
        // fragment of product i+8 is read while product i is multiplied: left to itself the compiler reads two fragments and
        // waits for them at once, which exposes the LDS latency 54 times per plane (0.75 ms instead of 0.5).
        const char* slots[3] = {ring + ((d + 2) % 3) * C0_PLANE_BYTES, ring + (d % 3) * C0_PLANE_BYTES,
                                ring + ((d + 1) % 3) * C0_PLANE_BYTES};  // planes d-1, d, d+1
        constexpr int NP = 108, DEPTH = 8;
        f16x8 fr[DEPTH];
        auto frag = [&](int i) {
            const int cg = i & 3, tap = i >> 2, kw = tap % 3, kh = (tap / 3) % 3, kd = tap / 9;
            return *reinterpret_cast<const f16x8*>(slots[kd] + fragk[kw] + (kh * C0_COLS + cg * 16) * 64);
        };
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) fr[i] = frag(i);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i >> 2], fr[i % DEPTH], acc[i & 3], 0, 0, 0);
            if (i + DEPTH < NP) fr[i % DEPTH] = frag(i + DEPTH);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) fin[cg] = acc[cg];
        __syncthreads();          // every wave is done with plane d-1: its slot takes plane d+2 next iteration
    }
    emit(dz1 - 1);
}

}  // namespace mvd

extern "C" {

size_t mvd_conv3d_f16_packed_weight_bytes(int Cin, int Cout) { return (Cin == 32 && Cout == 8) ? (size_t)27 * 64 * 16 : 0; }

int mvd_pack_conv3d_weights_f16(const float* w, int Cin, int Cout, void* packed, mvd_stream_t stream) {
    MVD_REQUIRE(w && packed, "pack_conv3d_weights_f16: NULL argument");
    MVD_REQUIRE(Cin == 32 && Cout == 8, "pack_conv3d_weights_f16: only the 32 -> 8 first layer of CostRegNet is built (got %d -> %d)", Cin, Cout);
    hipLaunchKernelGGL(mvd::pack_conv0_f16_kernel, dim3((27 * 64 * 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                       (_Float16*)packed);
    return mvd::launch_status("pack_conv3d_weights_f16");
}

int mvd_conv3d_bn_relu_f16in(const void* x, const void* packed_w, const float* scale, const float* shift, float* y, int B,
                             int D, int h, int w, int Cin, int Cout, int relu, mvd_stream_t stream) {
    MVD_REQUIRE(x && packed_w && scale && shift && y, "conv3d_f16in: NULL argument");
    MVD_REQUIRE(Cin == 32 && Cout == 8, "conv3d_f16in: only the 32 -> 8 first layer of CostRegNet is built (got %d -> %d)", Cin, Cout);
    MVD_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, "conv3d_f16in: non-positive dimension");
    MVD_REQUIRE((long long)h * w * 64 < 0x7fffffffLL, "conv3d_f16in: one input plane exceeds the 2 GiB offset range");
    mvd::Conv0Params p{};
    p.x = (const char*)x; p.wpk = (const char*)packed_w; p.scale = scale; p.shift = shift; p.y = y;
    p.B = B; p.D = D; p.h = h; p.w = w; p.relu = relu;
    p.tiles_x = (w + mvd::C0_TW - 1) / mvd::C0_TW;
    p.tiles_y = (h + mvd::C0_TH - 1) / mvd::C0_TH;
    // planes per workgroup: long marches amortise the two-plane ring fill, but the grid must still fill 256 CUs x 2
    const long long tiles = (long long)p.tiles_x * p.tiles_y;
    p.tiles_per_xcd = (int)((tiles + 7) / 8);
    int td = 32;
    while (td > 8 && tiles * B * ((D + td - 1) / td) < 2048) td /= 2;
    p.td = td;
    p.dgroups = (D + td - 1) / td;
    const long long nblk = 8LL * p.tiles_per_xcd * p.dgroups * B;
    MVD_REQUIRE(nblk <= 0x7fffffffLL, "conv3d_f16in: %lld workgroups exceed the grid limit", nblk);
    const size_t lds = 3 * (size_t)mvd::C0_PLANE_BYTES + 256 * 16;  // ring + the dump row of the idle staging lanes
    (void)hipFuncSetAttribute((const void*)mvd::conv0_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mvd::conv0_f16_kernel, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, p);
    return mvd::launch_status("conv3d_f16in");
}
}
