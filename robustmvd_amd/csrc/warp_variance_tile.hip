// K3, round-3 form ("tile"): the bilinear taps come from source FOOTPRINTS staged in LDS instead of 17 M gather
// instructions through the texture addresser (profiles/r02_k3_march_pmc.txt: TA 74 % busy, 56 % of a wave's life waiting).
// Same operator as warp_variance.hip — homo_warp (rmvd/models/blocks/utils.py:222-268) + the sum / sum of squares /
// variance of MVSNet.forward (rmvd/models/mvsnet.py:124-136) — same arithmetic operation for operation, results
// bit-identical to the gather kernels; C = 32, channel-last volume only.
//
// A workgroup (256 threads) owns a TW x TH = 32-pixel key tile and marches through `nch` <= 32 chunks of 8 depth planes.
//   PROBE   For every chunk and source view the tile's 8 corner samples (tile corners x first / last plane) give the
//           bounding box of the chunk's samples in the source map (convexity: header of warp_variance_exp.hip), kept one
//           cell wider on every side because an interior sample can round across a cell boundary no corner crosses.
//           Chunks whose boxes all hold at most WINPIX pixels go to pass 0, the others (near planes, where a sample moves
//           pixels per plane) to pass 1.  The boxes stay in LDS.
//   PASS 0  a sequence of UNITS u = (chunk, source view), ONE barrier per unit; in iteration u
//     C(u+1)  the box's rows (any shape, at most WINPIX pixels) are copied global -> LDS by LDS-DMA (`buffer_load_dwordx4 ...
//             lds`: no staging registers, 1 KiB per wave-instruction, bounds-checked by the buffer descriptor) into the
//             window unit u-1 used;
//     B(u)    blend: 8 lanes per pixel, 4 channels per lane; per plane one broadcast table read and four ds_read_b128
//             taps (LDS: 256 B/clk/CU against 64 B/clk/CU through the addresser), straight-line code, two planes in
//             flight, accumulators of the 8 planes in registers across the views; after the last view the chunk's
//             variance is stored;
//     L(u+1)  locate: thread t computes position, bilinear weights and 2x2 cell of exactly one (pixel, plane) of the unit
//             (32 x 8 = 256 of them, view wave-uniform -> transform through the scalar cache) and parks the weights and
//             the LDS address of the cell's first tap in the other table.  A cell outside the unit's box (not seen) marks
//             the chunk as failed: its stores are dropped and pass 1 redoes it.
//           A wave waits for global memory only at the barrier, for `vmcnt` of its own LDS-DMA (counted so that the chunk's 8
//           stores stay in flight).
//   PASS 1  the remaining chunks with the same table, taps gathered from the source map (buffer loads).
// The two passes are separate loops on purpose: one loop with both tap sources makes every accumulator a phi of two
// definitions, which this compiler does not coalesce (168 VGPRs + spills instead of 126).  LDS use: 2 windows + 10.6 KiB +
// 8 bytes per (chunk, view) of a workgroup's march.
#include "mvd_common.h"
#include <type_traits>
#include "warp_variance_common.h"

namespace mvd {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
// lane i receives lane i-K of its 16-lane row (lanes without a source keep their own value)
template <int K>
__device__ __forceinline__ unsigned row_shr(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x110 + K, 0xf, 0xf, false);
}

// One plane of one source view, 4 channels per lane as two explicit float pairs: bilinear blend of the cell's 4 taps, then
// the running sum and sum of squares (mvsnet.py:131-134).  The same fmaf chain per channel as accumulate_cell
// (warp_variance_common.h), so results are bit-identical; written on 2-vectors so that every operation IS one v_pk_fma_f32 /
// v_pk_add_f32 on consecutive registers (left to the SLP vectoriser, scalar code here gets paired ACROSS planes, with two
// v_mov per packed operation to assemble the operands: 145 moves per 96 packed operations).
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct Sums { f32x2 a1[2], a2[2]; };  // sum, sum of squares of channels (0,1), (2,3)
__device__ __forceinline__ void tap_pairs(const u32x4& t, f32x2& lo, f32x2& hi) {
    lo = f32x2{__uint_as_float(t.x), __uint_as_float(t.y)};
    hi = f32x2{__uint_as_float(t.z), __uint_as_float(t.w)};
}
__device__ __forceinline__ void tap_pairs(const u32x2& t, f32x2& lo, f32x2& hi) {  // fp16 taps: exact conversion
    const f16x2 l = as_h2(t.x), h = as_h2(t.y);
    lo = f32x2{(float)l.x, (float)l.y};
    hi = f32x2{(float)h.x, (float)h.y};
}
template <class TAP>
__device__ __forceinline__ void accumulate_cell_pk(Sums& s, const float4 w, const TAP (&t)[4]) {
    const float wk[4] = {w.x, w.y, w.z, w.w};
    f32x2 lo = {0.0f, 0.0f}, hi = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x2 tl, th;
        tap_pairs(t[k], tl, th);
        const f32x2 ws = {wk[k], wk[k]};
        lo = __builtin_elementwise_fma(tl, ws, lo);
        hi = __builtin_elementwise_fma(th, ws, hi);
    }
    s.a1[0] += lo; s.a1[1] += hi;
    s.a2[0] = __builtin_elementwise_fma(lo, lo, s.a2[0]);
    s.a2[1] = __builtin_elementwise_fma(hi, hi, s.a2[1]);
}

// LDS-DMA, one wave-instruction: 64 lanes x 16 bytes from `rsrc` + voff (per lane, bounds-checked) to LDS bytes
// [lds_dst, lds_dst + 1024) in lane order.  Inline assembly on purpose: behind the builtin the compiler makes the next LDS read
// of unknown aliasing wait for `vmcnt` of the copy (SIInsertWaitcnts), i.e. the blend of unit u would wait for the copy of
// unit u+1 it has just issued.  The kernel waits for its copies itself (MVD_UNIT_BARRIER).  M0 (the destination base) is written
// in the same statement that reads it and not restored: nothing else in this translation unit uses M0 (no LDS-DMA builtin, no
// s_movrel, no GWS; LDS instructions need no M0 on gfx9) -- tests/test_abi.py greps the kernel's ISA for other M0 uses.
__device__ __forceinline__ void lds_dma_b128(u32x4 rsrc, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}

#ifndef MVD_K3T_KO
#define MVD_K3T_KO 0  // knock-out builds (tools/ko_k3t.sh; WRONG results): 1 no LDS-DMA, 2 no tap reads, 4 no stores, 8 no locate
#endif

// LDS records written by the probe
struct UnitRec {          // one per (chunk, view) of the march: the unit's box in the source map, ready for C and L
    unsigned boxmin;      // first-tap cell range: xmin | ymin << 16 (padded coordinates)
    unsigned extent;      // (xmax - xmin) | (ymax - ymin) << 16
    unsigned pitch;       // window row pitch in bytes = nc * PIXB, nc = xmax - xmin + 2
    unsigned npix;        // nc * nr
    unsigned base;        // byte offset of the box's first pixel in the padded map
    unsigned skip;        // from the end of a window row to the start of the next row in the map: rowb - pitch
    float inv_nc;
    unsigned pad;
};
struct ViewRec {          // one per source view
    float M[12];          // composed transform (3 x 4)
    unsigned base_lo, base_hi;  // address of batch element b's padded map
    unsigned pad[2];
};

template <int TW, int WINPIX, bool F16, bool EXACT, int MINW, int SETS>
__global__ void __launch_bounds__(256, MINW) warp_variance_tile_kernel(WarpParams p, int nch) {
    constexpr int P = 8, NPX = 32;
    constexpr unsigned PIXB = F16 ? 64 : 128;  // bytes per pixel of the feature maps and of the volume
    constexpr unsigned QB = F16 ? 8 : 16;      // bytes per lane (4 channels)
    constexpr unsigned WINB = WINPIX * PIXB;   // bytes per LDS window
    constexpr int PPP = 1024 / PIXB;           // window pixels per LDS-DMA instruction (1 KiB)
    constexpr int LPX = PIXB / 16;             // DMA lanes per pixel
    constexpr int NPIECE = WINPIX / PPP;
    // LDS: 2 windows | 2 weight tables (256 x 16 B, [plane][pixel]) | 2 address tables (256 x 4 B, [pixel][plane]) |
    //      32 x 4 probe words | 32 fail words | V view records | nch x 8 depths | nch x V unit records
    constexpr unsigned TABW0 = 2 * WINB, TABO0 = TABW0 + 2 * 4096, PROBE0 = TABO0 + 2 * 1024, FAIL0 = PROBE0 + 512, VIEW0 = FAIL0 + 128;
    using TAP = typename std::conditional<F16, u32x2, u32x4>::type;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = p.h, w = p.w, D = p.D, V = p.V;
    const unsigned DEP0 = VIEW0 + (unsigned)V * sizeof(ViewRec), UNIT0 = DEP0 + (unsigned)nch * 32;

    // ---- decode the block index: xcd | (chunk group fastest, then tile within the XCD's band, then batch) ----
    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    const int dchunks = (D + P - 1) / P;
    const int dgroups = (dchunks + nch - 1) / nch;
    const int dg = j % dgroups; j /= dgroups;
    const int tile_in = j % p.tiles_per_xcd;
    const int b = j / p.tiles_per_xcd;
    const int tile = xcd * p.tiles_per_xcd + tile_in;
    if (tile >= p.tiles_x * p.tiles_y) return;  // block-uniform
    const int tyi = tile / p.tiles_x;
    const int x0 = (tile - tyi * p.tiles_x) * TW, y0 = tyi * (NPX / TW);
    // group dg marches through chunks dg, dg + dgroups, dg + 2 dgroups, ...: every group gets its share of the near chunks,
    // whose taps are gathered (pass 1, slower); contiguous ranges would leave all of them to group 0
    const int nchunks = (dchunks - dg + dgroups - 1) / dgroups;  // 1..nch
    auto chunk_of = [&](int jc) { return dg + jc * dgroups; };

    const int W2 = w + 3;
    const unsigned rowb = (unsigned)W2 * PIXB;            // bytes per padded row
    const unsigned img_bytes = (unsigned)(h + 3) * rowb;  // bytes per padded image

    // Per-thread constants (pixel coordinates, table addresses, byte offsets) are NOT kept in registers across the march:
    // every phase re-derives what it needs from an opaque copy of the thread index (a few integer operations per unit).
    // Hoisted out of the loops they would compete with the 64 accumulator registers and spill (scratch reloads are
    // vector-memory operations: each would drain the LDS-DMA and the stores in flight).
    auto opaque_tid = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t; };
    // blend role: 8 lanes per pixel, 4 channels per lane; pixel px = tid >> 3 of the tile at (px % TW, px / TW); ragged tiles
    // repeat the last column / row (same values stored to the same address)
    auto blend_pixel = [&](int t, int& bx, int& by) {
        const int px = t >> 3;
        bx = min(x0 + px % TW, w - 1);
        by = min(y0 + px / TW, h - 1);
    };

    float4 key;
    {
        int bx, by;
        blend_pixel(tid, bx, by);
        const char* kp = reinterpret_cast<const char*>(p.key) + (size_t)b * img_bytes + (unsigned)(by + 1) * rowb +
                         (unsigned)(bx + 1) * PIXB + (unsigned)(tid & 7) * QB;
        if constexpr (F16) {
            const u32x2 kh = *reinterpret_cast<const u32x2*>(kp);
            const f16x2 lo = as_h2(kh.x), hi = as_h2(kh.y);
            key = make_float4((float)lo.x, (float)lo.y, (float)hi.x, (float)hi.y);
        } else {
            key = *reinterpret_cast<const float4*>(kp);
        }
    }
    const float inv_nv = 1.0f / (float)(V + 1);  // mvsnet.py:135, V there counts the key view
    const size_t plane_bytes = (size_t)h * w * PIXB;

    auto src_rsrc = [&](int v) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.src.p[v]) + (size_t)b * img_bytes),
                                                 0, (int)img_bytes, 0x00020000);
    };

    // w/(w-1), h/(h-1) of the folded grid arithmetic: two IEEE divisions, done once and kept in scalar registers (left inside
    // `position` the compiler redoes them, 12 vector instructions each, for every unit)
    const float sx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)w / (float)(w - 1))));
    const float sy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)h / (float)(h - 1))));
    // sampling position of key pixel (fx, fy) at `depth` under the composed transform M (12 floats) of a source view,
    // clamped into the zero border (a sample outside the image lands on zero taps; v_med3_f32 sends NaN to -1); returns Z
    auto position = [&](const float (&M)[12], float fx, float fy, float depth, float& ix, float& iy) -> float {
        float Z;
        if constexpr (EXACT) {
            // the reference's own chain, one rounding per step: R @ (x*d, y*d, d) + T (utils.py:246-250), perspective divide
            // (IEEE), /((W-1)/2) - 1 (:256-257), then grid_sample's ((g+1)*W-1)/2
            const float half_w = (float)(w - 1) / 2.0f, half_h = (float)(h - 1) / 2.0f;
            const float gx = fx * depth, gy = fy * depth;
            const float X = ((M[0] * gx + M[1] * gy) + M[2] * depth) + M[3];
            const float Y = ((M[4] * gx + M[5] * gy) + M[6] * depth) + M[7];
            Z = ((M[8] * gx + M[9] * gy) + M[10] * depth) + M[11];
            ix = unnormalize_coord((X / Z) / half_w - 1.0f, (float)w);
            iy = unnormalize_coord((Y / Z) / half_h - 1.0f, (float)h);
        } else {
            // folded: ix = (X/Z) * w/(w-1) - 0.5 with 1/Z from v_rcp_f32 (within 1e-4 px of the chain above)
            const float ax = fmaf(M[0], fx, fmaf(M[1], fy, M[2]));
            const float ay = fmaf(M[4], fx, fmaf(M[5], fy, M[6]));
            const float az = fmaf(M[8], fx, fmaf(M[9], fy, M[10]));
            const float X = fmaf(ax, depth, M[3]), Y = fmaf(ay, depth, M[7]);
            Z = fmaf(az, depth, M[11]);
            const float rz = __builtin_amdgcn_rcpf(Z);
            ix = fmaf(X * rz, sx, -0.5f);
            iy = fmaf(Y * rz, sy, -0.5f);
        }
        ix = __builtin_amdgcn_fmed3f(ix, -1.0f, (float)w);
        iy = __builtin_amdgcn_fmed3f(iy, -1.0f, (float)h);
        return Z;
    };

    // ---- PROBE: per (chunk, view) the box of the tile's samples, from its 8 corner samples, one cell wider on every side
    // (an interior sample can round across a cell boundary that no corner crosses), as a ready-made unit record; the views'
    // transforms and the chunks' depths go to LDS as well (uniform reads in the march then cost neither scalar-load
    // latency nor SGPRs).  Bit j of lds_chunks: every view's box of the march's j-th chunk fits a window ----
    unsigned lds_chunks;
    {
        // thread = (view pv, corner): corners 0..3 of the tile at the chunk's first plane, 4..7 at its last
        const int pv = tid >> 3, co = tid & 7;
        const float cfx = (float)((co & 1) ? min(x0 + TW - 1, w - 1) : x0);
        const float cfy = (float)((co & 2) ? min(y0 + NPX / TW - 1, h - 1) : y0);
        float M[12];
        {
            const float* Mg = p.M + ((size_t)min(pv, V - 1) * p.B + b) * 12;
#pragma unroll
            for (int k = 0; k < 12; ++k) M[k] = Mg[k];
        }
        if (co == 0 && pv < V) {
            ViewRec* vr = reinterpret_cast<ViewRec*>(lds + VIEW0) + pv;
#pragma unroll
            for (int k = 0; k < 12; ++k) vr->M[k] = M[k];
            const unsigned long long sb = (unsigned long long)(reinterpret_cast<const char*>(p.src.p[pv]) + (size_t)b * img_bytes);
            vr->base_lo = (unsigned)sb; vr->base_hi = (unsigned)(sb >> 32);
        }
        if (tid < 32) *reinterpret_cast<unsigned*>(lds + FAIL0 + tid * 4) = 0u;
        const float* dg_ = p.depth + (size_t)b * D;
        for (int e = tid; e < nchunks * P; e += 256) *reinterpret_cast<float*>(lds + DEP0 + e * 4) = dg_[min(chunk_of(e / P) * P + e % P, D - 1)];
        __syncthreads();
        for (int jc = 0; jc < nchunks; ++jc) {
            const float depth = *reinterpret_cast<const float*>(lds + DEP0 + (jc * P + ((co & 4) ? P - 1 : 0)) * 4);
            float ix, iy;
            const float Z = position(M, cfx, cfy, depth, ix, iy);
            const unsigned cell = ((unsigned)((int)floorf(iy) + 1) << 16) | (unsigned)((int)floorf(ix) + 1);
            const bool ok = Z > 0.0f;  // false for NaN too: then the box is the whole map (does not fit)
            unsigned mn = ok ? cell : 0u, mx = ok ? cell : 0xffffffffu;
            mn = pk_min_u16(mn, row_shr<1>(mn)); mx = pk_max_u16(mx, row_shr<1>(mx));
            mn = pk_min_u16(mn, row_shr<2>(mn)); mx = pk_max_u16(mx, row_shr<2>(mx));
            mn = pk_min_u16(mn, row_shr<4>(mn)); mx = pk_max_u16(mx, row_shr<4>(mx));
            // lane 7 of every 8-lane group holds its view's box of first-tap cells (padded coordinates): widen, clamp to the
            // cells a clamped sample can have (columns 0..w+1, rows 0..h+1)
            const int xmin = max((int)(mn & 0xffffu) - 1, 0), ymin = max((int)(mn >> 16) - 1, 0);
            const int xmax = min((int)(mx & 0xffffu) + 1, w + 1), ymax = min((int)(mx >> 16) + 1, h + 1);
            const int nc = xmax - xmin + 2, nr = ymax - ymin + 2;  // +1: the cell's second column / row
            const bool mine = (co == 7) && (pv < V);
            if (mine) {
                UnitRec r;
                r.boxmin = (unsigned)xmin | ((unsigned)ymin << 16);
                r.extent = (unsigned)(xmax - xmin) | ((unsigned)(ymax - ymin) << 16);
                r.pitch = (unsigned)nc * PIXB;
                r.npix = (unsigned)(nc * nr);
                r.base = (unsigned)(ymin * W2 + xmin) * PIXB;
                r.skip = rowb - (unsigned)nc * PIXB;
                r.inv_nc = 1.0f / (float)nc;
                r.pad = 0;
                *(reinterpret_cast<UnitRec*>(lds + UNIT0) + jc * V + pv) = r;
            }
            const bool any_bad = __builtin_amdgcn_ballot_w64(mine && nc * nr > WINPIX) != 0;
            if ((tid & 63) == 0) *reinterpret_cast<unsigned*>(lds + PROBE0 + (jc * 4 + wv) * 4) = any_bad ? 1u : 0u;
        }
        __syncthreads();
        unsigned m = 0;
        for (int jc = 0; jc < nchunks; ++jc) {
            const u32x4 f = *reinterpret_cast<const u32x4*>(lds + PROBE0 + jc * 16);
            if ((f.x | f.y | f.z | f.w) == 0) m |= 1u << jc;
        }
        lds_chunks = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
        if constexpr ((MVD_K3T_KO & 16) != 0) lds_chunks = 0;
    }

    // ---- L: locate one (pixel, plane) of unit (the march's chunk jc, view v) per thread into table `slot` ----
    // wb >= 0 (pass 0): the table gets the LDS address of the cell's first tap in window wb, laid out by the unit's box; a
    // cell outside the box (never seen; the probe's box is one cell wider than the corners' on every side) fails the chunk.
    // wb < 0 (pass 1): the table gets the cell's byte offset in the source map.
    auto locate = [&](int jc, int v, int slot, int wb) {
        if constexpr ((MVD_K3T_KO & 8) != 0) return;
        // thread = (pixel lpx of the tile, plane li = tid >> 5 of the chunk)
        const int t = opaque_tid(), lpx = t & 31, li = t >> 5;
        const float lfx = (float)min(x0 + lpx % TW, w - 1), lfy = (float)min(y0 + lpx / TW, h - 1);
        const float depth = *reinterpret_cast<const float*>(lds + DEP0 + (jc * P + li) * 4);
        const ViewRec* vr = reinterpret_cast<const ViewRec*>(lds + VIEW0) + v;
        const float4 m0 = *reinterpret_cast<const float4*>(vr->M), m1 = *reinterpret_cast<const float4*>(vr->M + 4),
                     m2 = *reinterpret_cast<const float4*>(vr->M + 8);
        const float M[12] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w};
        float ix, iy;
        position(M, lfx, lfy, depth, ix, iy);
        const float xf = floorf(ix), yf = floorf(iy);
        const float wx = ix - xf, wy = iy - yf;
        const float ux = 1.0f - wx, uy = 1.0f - wy;
        *reinterpret_cast<float4*>(lds + TABW0 + slot * 4096 + t * 16) = make_float4(ux * uy, wx * uy, ux * wy, wx * wy);
        unsigned* const ent = reinterpret_cast<unsigned*>(lds + TABO0 + slot * 1024 + (lpx * P + li) * 4);
        const int cx = (int)xf + 1, cy = (int)yf + 1;  // padded coordinates of the cell's first tap
        if (wb < 0) {  // wave-uniform
            *ent = (unsigned)(cy * W2 + cx) * PIXB;
        } else {
            const u32x4 rec = *reinterpret_cast<const u32x4*>(lds + UNIT0 + (jc * V + v) * sizeof(UnitRec));  // boxmin, extent, pitch, npix
            const unsigned dx = (unsigned)cx - (rec.x & 0xffffu), dy = (unsigned)cy - (rec.x >> 16);
            *ent = (unsigned)wb * WINB + dy * rec.z + dx * PIXB;
            const bool outside = dx > (rec.y & 0xffffu) || dy > (rec.y >> 16);
            if (__builtin_amdgcn_ballot_w64(outside) != 0) {  // wave-uniform, never taken in practice
                if ((t & 63) == 0) *reinterpret_cast<unsigned*>(lds + FAIL0 + jc * 4) = 1u;
            }
        }
    };

    // ---- C: copy the footprint (box) of unit (chunk jc, view v) into window `wb`: rows of the box, global -> LDS by LDS-DMA ----
    auto copy_window = [&](int jc, int v, int wb) {
        if constexpr ((MVD_K3T_KO & 1) != 0) return;
        const char* rp = lds + UNIT0 + (jc * V + v) * sizeof(UnitRec);
        const u32x4 ra = *reinterpret_cast<const u32x4*>(rp), rb = *reinterpret_cast<const u32x4*>(rp + 16);  // uniform reads
        const u32x2 sb = *reinterpret_cast<const u32x2*>(lds + VIEW0 + v * sizeof(ViewRec) + 48);
        // raw buffer descriptor: base, stride 0, num_records = bytes of the padded map, raw-buffer flags
        const u32x4 rsrc = {(unsigned)__builtin_amdgcn_readfirstlane((int)sb.x), (unsigned)__builtin_amdgcn_readfirstlane((int)sb.y) & 0xffffu,
                            img_bytes, 0x00020000u};
        const int npix = __builtin_amdgcn_readfirstlane((int)ra.w);
        const int ln = opaque_tid() & 63;
        const float inv_nc = __uint_as_float(rb.z);
        const unsigned base = rb.x + (unsigned)(ln % LPX) * 16u + (unsigned)(ln / LPX) * PIXB;  // + this lane's place in piece 0
        const float nf0 = (float)(ln / LPX) + 0.5f;
#pragma unroll
        for (int k = 0; k < (NPIECE + 3) / 4; ++k) {
            const int piece = wv + 4 * k;
            if (piece * PPP < npix) {  // wave-uniform (npix <= WINPIX in pass 0: never past the window)
                // window pixel n = piece * PPP + lane / LPX sits in box row r = n / nc (exact for these magnitudes).  Lanes
                // past the box's last row read past the footprint: inside the map (unused), or beyond its end, where the
                // descriptor's bounds check returns zeros
                const int r = (int)((nf0 + (float)(piece * PPP)) * inv_nc);
                lds_dma_b128(rsrc, base + (unsigned)(piece * 1024) + (unsigned)r * rb.y, (unsigned)(wb * WINB + piece * 1024));
            }
        }
    };

    // ---- B: blend the unit (chunk jc, view v) in table `slot`; LDSTAPS: taps from window `slot`, else gathered from the
    // source map.  Software-pipelined by hand, two planes in flight: the taps of plane i+2 are requested when plane i's
    // registers are free (sched_barrier: left alone, the scheduler hoists the reads of all 8 planes and spills).
    auto blend = [&](Sums (&sm)[P], int jc, int v, int slot, auto lds_tag) {
        constexpr bool LDSTAPS = decltype(lds_tag)::value;
        const int t = opaque_tid();
        const unsigned q_b = (unsigned)(t & 7) * QB;
        const char* tw = lds + TABW0 + slot * 4096 + (t >> 3) * 16;
        // the 8 planes' addresses of this lane's pixel: two reads up front, off the per-plane dependency chain
        const char* to = lds + TABO0 + slot * 1024 + (t >> 3) * 32;
        const u32x4 oa = *reinterpret_cast<const u32x4*>(to), ob = *reinterpret_cast<const u32x4*>(to + 16);
        const unsigned ofs[P] = {oa.x, oa.y, oa.z, oa.w, ob.x, ob.y, ob.z, ob.w};
        unsigned pitch = 0;
        __amdgpu_buffer_rsrc_t rsrc;
        if constexpr (LDSTAPS) pitch = *reinterpret_cast<const unsigned*>(lds + UNIT0 + (jc * V + v) * sizeof(UnitRec) + 8);
        else rsrc = src_rsrc(v);
        auto weights = [&](int i) { return *reinterpret_cast<const float4*>(tw + i * (NPX * 16)); };
        auto taps = [&](int i, TAP (&f)[4]) {
            const unsigned o = ofs[i] + q_b;
            if constexpr ((MVD_K3T_KO & 2) != 0) {
                if constexpr (F16) { f[0] = u32x2{o, o + 1}; f[1] = u32x2{o + 2, o + 3}; f[2] = u32x2{o + pitch, o + 5}; f[3] = u32x2{o + 6, o + 7}; }
                else { f[0] = u32x4{o, o + 1, o + 2, o + 3}; f[1] = u32x4{o + 4, o + 5, o + 6, o + 7};
                       f[2] = u32x4{o + pitch, o + 9, o + 10, o + 11}; f[3] = u32x4{o + 12, o + 13, o + 14, o + 15}; }
            } else if constexpr (LDSTAPS) {
                f[0] = *reinterpret_cast<const TAP*>(lds + o);
                f[1] = *reinterpret_cast<const TAP*>(lds + o + PIXB);
                f[2] = *reinterpret_cast<const TAP*>(lds + o + pitch);
                f[3] = *reinterpret_cast<const TAP*>(lds + o + pitch + PIXB);
            } else if constexpr (F16) {
                f[0] = load_b64(rsrc, o, 0); f[1] = load_b64(rsrc, o + PIXB, 0);
                f[2] = load_b64(rsrc, o, rowb); f[3] = load_b64(rsrc, o + PIXB, rowb);
            } else {
                f[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);
                f[1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + PIXB, 0, 0);
                f[2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, rowb, 0);
                f[3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + PIXB, rowb, 0);
            }
        };
        // Sweep coherence: between planes a sample moves a fraction of a pixel, so the 2x2 cell of plane i is often the cell of
        // plane i-2 for every pixel of the wave (host count at the headline poses: 54 % of the tap fetches remain).  Plane i
        // uses tap set i & 1; the set is re-read only when some lane's cell differs (wave-uniform branch, registers refreshed
        // in place: no per-lane select, and the accumulators are not involved in the branch).
        // `done` = the plane just accumulated from this tap set.  The empty asm ties the refresh (through the address it tests) to
        // the finished sums: without it the compiler sinks all eight accumulations below all the conditional refreshes and
        // keeps eight tap sets alive (copies + spills).
        auto next = [&](int i, float4& wq, TAP (&f)[4], Sums& done) {
            wq = weights(i);
            unsigned oi = ofs[i];
            asm volatile("" : "+v"(done.a1[0]), "+v"(done.a1[1]), "+v"(done.a2[0]), "+v"(done.a2[1]), "+v"(oi));
            if (__builtin_amdgcn_ballot_w64(oi != ofs[i - SETS]) != 0) {
                const unsigned o = oi + q_b;
                if constexpr ((MVD_K3T_KO & 2) != 0) { taps(i, f); }
                else if constexpr (LDSTAPS) {
                    f[0] = *reinterpret_cast<const TAP*>(lds + o);
                    f[1] = *reinterpret_cast<const TAP*>(lds + o + PIXB);
                    f[2] = *reinterpret_cast<const TAP*>(lds + o + pitch);
                    f[3] = *reinterpret_cast<const TAP*>(lds + o + pitch + PIXB);
                } else if constexpr (F16) {
                    f[0] = load_b64(rsrc, o, 0); f[1] = load_b64(rsrc, o + PIXB, 0);
                    f[2] = load_b64(rsrc, o, rowb); f[3] = load_b64(rsrc, o + PIXB, rowb);
                } else {
                    f[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);
                    f[1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + PIXB, 0, 0);
                    f[2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, rowb, 0);
                    f[3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + PIXB, rowb, 0);
                }
            }
        };
        if constexpr (SETS == 2) {
            float4 wA = weights(0), wB = weights(1);
            TAP fA[4], fB[4];
            taps(0, fA);
            taps(1, fB);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < P; i += 2) {
                accumulate_cell_pk(sm[i], wA, fA);
                __builtin_amdgcn_sched_barrier(0);
                if (i + 2 < P) next(i + 2, wA, fA, sm[i]);
                __builtin_amdgcn_sched_barrier(0);
                accumulate_cell_pk(sm[i + 1], wB, fB);
                __builtin_amdgcn_sched_barrier(0);
                if (i + 3 < P) next(i + 3, wB, fB, sm[i + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // One tap set (16 registers fewer: 128 VGPRs, four waves per SIMD, which cover the LDS latency of a refresh instead
            // of a second set); a plane re-reads its taps only when some lane's cell differs from the previous plane's (host
            // count: 38 % of the fetches remain)
            float4 wA = weights(0);
            TAP fA[4];
            taps(0, fA);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < P; ++i) {
                accumulate_cell_pk(sm[i], wA, fA);
                __builtin_amdgcn_sched_barrier(0);
                if (i + 1 < P) next(i + 1, wA, fA, sm[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    auto reset_sums = [&](Sums (&sm)[P]) {
        const f32x2 klo = {key.x, key.y}, khi = {key.z, key.w};
#pragma unroll
        for (int i = 0; i < P; ++i) { sm[i].a1[0] = klo; sm[i].a1[1] = khi; sm[i].a2[0] = klo * klo; sm[i].a2[1] = khi * khi; }
    };

    // ---- S: variance of chunk c, always 8 stores per lane: planes past D, or every plane when `drop` (a failed chunk of
    // pass 0), go through an empty descriptor (dropped, but counted by vmcnt) ----
    float amax = 0.f;  // max |variance| this lane has stored (by-product for the next layer's range scaling), if asked for
    const bool want_absmax = p.absmax != nullptr;
    auto store_chunk = [&](const Sums (&sm)[P], int c, bool drop) {
        int bx, by;
        const int t = opaque_tid();
        blend_pixel(t, bx, by);
        const unsigned out_off = ((unsigned)by * (unsigned)w + (unsigned)bx) * PIXB + (unsigned)(t & 7) * QB;
        // one descriptor per plane: base advanced by scalar adds, num_records 0 for a plane to be dropped
        const unsigned long long plane0 = (unsigned long long)(reinterpret_cast<char*>(p.out) + ((size_t)b * D + (size_t)c * P) * plane_bytes);
        const int nvalid = drop ? 0 : min(P, D - c * P);  // planes of this chunk that exist (and are kept)
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const float mx = sm[i].a1[0].x * inv_nv, my = sm[i].a1[0].y * inv_nv, mz = sm[i].a1[1].x * inv_nv, mw = sm[i].a1[1].y * inv_nv;
            const float4 r = make_float4(fmaf(sm[i].a2[0].x, inv_nv, -mx * mx), fmaf(sm[i].a2[0].y, inv_nv, -my * my),
                                         fmaf(sm[i].a2[1].x, inv_nv, -mz * mz), fmaf(sm[i].a2[1].y, inv_nv, -mw * mw));
            const unsigned long long pb = plane0 + (i < nvalid ? (unsigned long long)i * plane_bytes : 0ull);
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(pb), 0, i < nvalid ? (int)plane_bytes : 0, 0x00020000);
            if constexpr ((MVD_K3T_KO & 4) != 0) { if (r.x != 123.456f) continue; }
            if (want_absmax && i < nvalid) {  // wave-uniform.  Finite values only: a quad with an inf or NaN is left out
                const float qm = fmaxf(fmaxf(fabsf(r.x), fabsf(r.y)), fmaxf(fabsf(r.z), fabsf(r.w)));  // v_max_f32 drops NaNs
                const bool fin = fabsf(r.x) + fabsf(r.y) + fabsf(r.z) + fabsf(r.w) <= 3.402823466e38f;   // false if any is inf / NaN
                amax = fin ? fmaxf(amax, qm) : amax;
            }
            if constexpr (F16) {  // round to nearest even, one rounding
                const f16x2 lo = {(_Float16)r.x, (_Float16)r.y}, hi = {(_Float16)r.z, (_Float16)r.w};
                store_b64(u32x2{as_u32(lo), as_u32(hi)}, orsrc, out_off);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(r.x), __float_as_uint(r.y), __float_as_uint(r.z), __float_as_uint(r.w)},
                                                       orsrc, out_off, 0, 0);
            }
        }
    };

    // one barrier per unit: LDS traffic of this wave complete (tables), its LDS-DMA landed; `NST` younger stores may stay in flight
#define MVD_UNIT_BARRIER(NST) asm volatile("s_waitcnt vmcnt(" #NST ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
    // units of a pass: chunk jc (bit index in `mask`) x view v; jc = -1 past the end
    auto first_chunk = [](unsigned mask) { return mask ? (int)__builtin_ctz(mask) : -1; };
    auto advance = [&](unsigned mask, int& jc, int& v) {
        if (++v < V) return;
        v = 0;
        const unsigned rest = jc >= 31 ? 0u : (mask >> (jc + 1));
        jc = rest ? jc + 1 + (int)__builtin_ctz(rest) : -1;
    };
    std::integral_constant<bool, true> yes;
    std::integral_constant<bool, false> no;

    // ---- PASS 0: chunks whose taps come from LDS windows.  Per unit: copy of the next unit issued, this unit blended (its
    // chunk stored after the last view), next unit located, barrier ----
    if (lds_chunks) {
        const unsigned mask = lds_chunks;
        int jb = first_chunk(mask);
        copy_window(jb, 0, 0);
        locate(jb, 0, 0, 0);
        MVD_UNIT_BARRIER(0);
        int slot = 0;  // table and window of the unit being blended
        while (jb >= 0) {
            const unsigned rest = jb >= 31 ? 0u : (mask >> (jb + 1));
            const int jn = rest ? jb + 1 + (int)__builtin_ctz(rest) : -1;  // the next chunk of this pass
            Sums sm[P];
            reset_sums(sm);
            for (int v = 0; v + 1 < V; ++v) {
                copy_window(jb, v + 1, slot ^ 1);
                blend(sm, jb, v, slot, yes);
                locate(jb, v + 1, slot ^ 1, slot ^ 1);
                MVD_UNIT_BARRIER(0);
                slot ^= 1;
            }
            if (jn >= 0) copy_window(jn, 0, slot ^ 1);
            blend(sm, jb, V - 1, slot, yes);
            {
                const bool drop = __builtin_amdgcn_readfirstlane((int)*reinterpret_cast<const unsigned*>(lds + FAIL0 + jb * 4)) != 0;  // set before the previous barrier at the latest
                store_chunk(sm, chunk_of(jb), drop);
            }
            if (jn >= 0) locate(jn, 0, slot ^ 1, slot ^ 1);
            MVD_UNIT_BARRIER(8);
            slot ^= 1;
            jb = jn;
        }
    }

    // ---- PASS 1: the other chunks (and the failed ones), taps gathered from the source maps ----
    unsigned failed = 0;
    for (int jc = 0; jc < nchunks; ++jc)
        if (*reinterpret_cast<const unsigned*>(lds + FAIL0 + jc * 4) != 0) failed |= 1u << jc;
    failed = (unsigned)__builtin_amdgcn_readfirstlane((int)failed);
    const unsigned all = nchunks >= 32 ? 0xffffffffu : ((1u << nchunks) - 1u);
    const unsigned gmask = (all & ~lds_chunks) | failed;
    if (gmask) {
        int jl = first_chunk(gmask), vl = 0;
        locate(jl, vl, 0, -1); advance(gmask, jl, vl);
        int jb = first_chunk(gmask), vb = 0, slot = 0;
        Sums sm[P];
        reset_sums(sm);
        while (jb >= 0) {
            // table `slot` complete; every wave is done reading table `slot ^ 1`.  vmcnt(8): the previous chunk's stores stay
            // in flight (the gathers of the last blend have been consumed)
            MVD_UNIT_BARRIER(8);
            if (jl >= 0) { locate(jl, vl, slot ^ 1, -1); advance(gmask, jl, vl); }
            blend(sm, jb, vb, slot, no);
            if (vb + 1 == V) {
                store_chunk(sm, chunk_of(jb), false);
                reset_sums(sm);
            }
            slot ^= 1;
            advance(gmask, jb, vb);
        }
    }
#undef MVD_UNIT_BARRIER
    if (want_absmax) {  // non-negative floats order like their bit patterns; one global atomic per workgroup (atomics on one
        // address serialise), the four waves meet in an LDS word that no later code reads
        unsigned m = __float_as_uint(amax);
        m = max(m, row_shr<1>(m)); m = max(m, row_shr<2>(m)); m = max(m, row_shr<4>(m)); m = max(m, row_shr<8>(m));
        m = max(m, (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x142, 0xa, 0xf, false));
        m = max(m, (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x143, 0xc, 0xf, false));
        unsigned* wm = reinterpret_cast<unsigned*>(lds + PROBE0);
        __syncthreads();  // every wave is past its last use of the probe words
        if ((tid & 63) == 63) wm[wv] = m;
        __syncthreads();
        if (tid == 0) raise_absmax(p.absmax, __uint_as_float(max(max(wm[0], wm[1]), max(wm[2], wm[3]))));
    }
}

static size_t tile_lds_bytes(int win, bool f16, int V, int nch) {
    return 2 * (size_t)win * (f16 ? 64 : 128) + 2 * 4096 + 2 * 1024 + 512 + 128 + (size_t)V * sizeof(ViewRec) + (size_t)nch * 32 +
           (size_t)nch * V * sizeof(UnitRec);
}

template <int TW, int WINPIX, bool F16, bool EXACT, int MINW, int SETS>
static int launch_tile_variant(const WarpParams& p, const dim3& grid, hipStream_t st, int nch) {
    const size_t lds = tile_lds_bytes(WINPIX, F16, p.V, nch);
    auto* fn = warp_variance_tile_kernel<TW, WINPIX, F16, EXACT, MINW, SETS>;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return launch_status("warp_variance_tile: hipFuncSetAttribute");
    }
    hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, p, nch);
    return MVD_OK;
}

// tile-kernel launcher: tw = tile width (8, 16 or 32; height 32 / tw), win = LDS window pixels, nch = chunks of 8 planes per
// workgroup (at most; capped so that the workgroup's LDS keeps the variant's occupancy), sets = tap register sets (2: three
// waves per SIMD, 1: four).  warp_tile_supported() says whether the shape can use this kernel at all.
bool warp_tile_supported(const WarpParams& p, bool f16) {
    return p.h + 3 < 65536 && p.w + 3 < 65536 && (long long)p.h * p.w * (f16 ? 64 : 128) < 0x7fffffffLL && p.V <= 32;
}

int launch_warp_tile(const WarpParams& p0, hipStream_t st, int tw, int win, int nch, int sets, bool f16, bool exact) {
    WarpParams p = p0;
    const int th = 32 / tw;
    p.tiles_x = (p.w + tw - 1) / tw;
    p.tiles_y = (p.h + th - 1) / th;
    const long long tiles = (long long)p.tiles_x * p.tiles_y;
    p.tiles_per_xcd = (int)((tiles + 7) / 8);
    const int dchunks = (p.D + 7) / 8;
    // one mode bit per chunk in a 32-bit mask; 32 bytes of LDS per chunk and per (chunk, view): as many chunks as keep
    // 4 (one tap set) or 3 (two sets) workgroups per CU
    nch = nch < 1 ? 1 : (nch > 32 ? 32 : nch);
    const size_t budget = (size_t)160 * 1024 / (sets == 1 ? 4 : 3);
    while (nch > 1 && tile_lds_bytes(win, f16, p.V, nch) > budget) --nch;
    const int dgroups = (dchunks + nch - 1) / nch;
    const long long nblk = 8LL * p.tiles_per_xcd * dgroups * p.B;
    if (nblk > 0x7fffffffLL) {
        set_error("warp_variance: %lld workgroups exceed the grid limit", nblk);
        return MVD_ERR_INVALID_ARG;
    }
    const dim3 grid((unsigned)nblk);
    int rc = MVD_ERR_INVALID_ARG;
    timing_begin(st);
#define MVD_T(TW_, WIN_, MW_, SETS_)                                                                                   \
    if (tw == TW_ && win == WIN_ && sets == SETS_) {                                                                  \
        rc = f16 ? (exact ? launch_tile_variant<TW_, WIN_, true, true, MW_, SETS_>(p, grid, st, nch)                  \
                          : launch_tile_variant<TW_, WIN_, true, false, MW_, SETS_>(p, grid, st, nch))                \
                 : (exact ? launch_tile_variant<TW_, WIN_, false, true, MW_, SETS_>(p, grid, st, nch)                 \
                          : launch_tile_variant<TW_, WIN_, false, false, MW_, SETS_>(p, grid, st, nch));              \
    }
    MVD_T(8, 104, 4, 1) MVD_T(8, 128, 4, 1)  // product: fp32 maps, fp16 maps (half the window bytes)
#ifdef MVD_EXPERIMENTS
    MVD_T(8, 128, 3, 2) MVD_T(8, 96, 4, 1) MVD_T(16, 128, 3, 2) MVD_T(16, 104, 4, 1) MVD_T(16, 96, 4, 1) MVD_T(32, 128, 3, 2) MVD_T(32, 104, 4, 1)
#endif
#undef MVD_T
    timing_end(st);
    if (rc == MVD_ERR_INVALID_ARG) {
        set_error("warp_variance: tile variant tw=%d win=%d sets=%d is not compiled", tw, win, sets);
        return rc;
    }
    if (rc) return rc;
    return launch_status("warp_variance_tile");
}

}  // namespace mvd
