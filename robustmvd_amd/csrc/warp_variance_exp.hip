// Experimental forms of K3 (bit-identical to the product kernel, none faster; DESIGN.md section 4 records the
// measurements).  NOT part of libmvd_hip.so: compiled only into robustmvd_amd/lib_exp/libmvd_hip_exp.so
// (`make exp`, -DMVD_EXPERIMENTS), which tools/ and tests/test_hip_shapes.py::test_warp_variance_experimental_
// variants_match_default load explicitly.
#include "warp_variance_common.h"

namespace mvd {

// Footprint copy of the LDS-staged kernel, global -> LDS directly (LDS-DMA, no staging registers): wave r moves box
// rows r, r+4, ...; one instruction writes 64 consecutive float4s (1 KiB) of an LDS row, so rows are pitched to
// LDS_ROWQ float4s and a row tail that overshoots `rowq` lands in the row's own padding.
constexpr int LDS_ROWQ = 128;  // float4s per LDS row = 16 pixels of 32 channels
template <int NROW, int NCOL>
__device__ __forceinline__ void lds_dma_copy(float4* __restrict__ dst, const float4* __restrict__ g, int pitchq, int rowq,
                                             int rh, int wave, int lane) {
#pragma unroll
    for (int r = 0; r < NROW; ++r) {
        const int row = wave + 4 * r;
        if (row < rh) {  // wave-uniform
#pragma unroll
            for (int c = 0; c < NCOL; ++c)
                if (c * 64 < rowq)  // wave-uniform
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(g + (size_t)row * pitchq + min(lane + 64 * c, rowq - 1)),
                        (__attribute__((address_space(3))) void*)(dst + row * LDS_ROWQ + c * 64), 16, 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-staged form (C = 32, channel-last output): taps come from LDS (256 B/clk/CU) instead of the L1 /
// texture-addresser path (64 B/clk/CU), which is what bounds the direct kernel above (TA 84 % busy).
//
// A workgroup owns an 8x8 key tile for ND consecutive planes.  Per source view it copies the tile's source
// FOOTPRINT — the axis-aligned bounding box of all its samples in the zero-bordered image — into one of two
// LDS buffers with coalesced row loads, and takes every bilinear tap from there with ds_read_b128.  The copy
// of view v+1 is issued (global -> registers) before the taps of view v are computed and written to the other
// buffer afterwards: one barrier per view, the L2 latency of the copy hidden under a view's worth of work.
//
// Footprint bound: for a fixed plane the homography maps the tile to a convex quad, and for a fixed pixel the
// sample moves monotonically along its epipolar line with depth, so — as long as Z > 0 at the 8 corners (tile
// corners x first/last plane), which bounds Z > 0 in between because Z is multilinear in (x, y, d) — every
// (clamped) sample lies in the bounding box of the 8 clamped corner samples.  A box that does not fit the LDS
// buffer, or a corner with Z <= 0, sends that (tile, chunk, view) down the direct-gather path (block-uniform).
// Tap coordinates are clamped into the staged box, so a sample pushed across a box edge by rounding
// (weight < 1e-5) still reads valid LDS.  Results are bit-identical to the direct kernel.
template <int ND>
__global__ void __launch_bounds__(256, 2) warp_variance_lds_kernel(WarpParams p) {
    constexpr int C = 32, Q = 8, TX = 8, TY = 4;
    constexpr unsigned PIX = 128;
    constexpr int RH_MAX = 12, ROWQ_MAX = LDS_ROWQ;  // boxes up to 12 rows x 16 pixels: 24 KiB per buffer
    constexpr int NROW = RH_MAX / 4, NCOL = ROWQ_MAX / 64;
    __shared__ float4 region[2][RH_MAX * LDS_ROWQ];  // 48 KiB -> 3 workgroups per CU

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int q = tid & 7;
    const int slot = tid >> 3;  // 0..31: one pixel of the 4x8 tile per 8 lanes
    const int lx = slot & 7, ly0 = slot >> 3;
    const int h = p.h, w = p.w, D = p.D;

    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    const int dchunks = (D + ND - 1) / ND;
    const int dc = j % dchunks; j /= dchunks;
    const int tile_in = j % p.tiles_per_xcd;
    const int b = j / p.tiles_per_xcd;
    const int tile = xcd * p.tiles_per_xcd + tile_in;
    if (tile >= p.tiles_x * p.tiles_y) return;  // block-uniform
    const int tyi = tile / p.tiles_x;
    const int x0 = (tile - tyi * p.tiles_x) * TX, y0 = tyi * TY;
    const int d0 = dc * ND;
    const int dl = min(d0 + ND, D) - 1;

    const float sx = (float)w / (float)(w - 1), sy = (float)h / (float)(h - 1);
    const float xhi = (float)w, yhi = (float)h;
    const int W2 = w + 3;
    const unsigned rowb = (unsigned)W2 * PIX;
    const unsigned img_bytes = (unsigned)(h + 3) * rowb;
    const int xs = min(x0 + lx, w - 1);
    const int ys[1] = {min(y0 + ly0, h - 1)};
    const float fxs = (float)xs;
    const float* __restrict__ dvals = p.depth + (size_t)b * D;
    float dep[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) dep[i] = dvals[min(d0 + i, D - 1)];
    const float dfirst = dvals[d0], dlast = dvals[dl];
    const float cxs[2] = {(float)x0, (float)min(x0 + TX - 1, w - 1)};
    const float cys[2] = {(float)y0, (float)min(y0 + TY - 1, h - 1)};

    float4 s1[1][ND], s2[1][ND];
#pragma unroll
    for (int pi = 0; pi < 1; ++pi) {
        const float4 k = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.key) + (size_t)b * img_bytes +
                                                          (unsigned)(ys[pi] + 1) * rowb + (unsigned)(xs + 1) * PIX + q * 16);
        const float4 k2 = make_float4(k.x * k.x, k.y * k.y, k.z * k.z, k.w * k.w);
#pragma unroll
        for (int i = 0; i < ND; ++i) { s1[pi][i] = k; s2[pi][i] = k2; }
    }

    // footprint box of one view in PADDED pixel coordinates (x+1, y+1); staged == it fits the LDS buffer
    struct Box { int x0, y0, rw, rh; bool staged; };
    auto footprint = [&](int v) {
        const float* __restrict__ M = p.M + ((size_t)v * p.B + b) * 12;
        float bx0 = 3e38f, bx1 = -3e38f, by0 = 3e38f, by1 = -3e38f, zmin = 3e38f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float cx = cxs[a & 1], cy = cys[a >> 1];
            const float ax = fmaf(M[0], cx, fmaf(M[1], cy, M[2])), ay = fmaf(M[4], cx, fmaf(M[5], cy, M[6]));
            const float az = fmaf(M[8], cx, fmaf(M[9], cy, M[10]));
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float cd = e ? dlast : dfirst;
                const float X = fmaf(ax, cd, M[3]), Y = fmaf(ay, cd, M[7]), Z = fmaf(az, cd, M[11]);
                const float rz = __builtin_amdgcn_rcpf(Z);
                const float ix = fminf(fmaxf(fmaf(X * rz, sx, -0.5f), -1.0f), xhi);
                const float iy = fminf(fmaxf(fmaf(Y * rz, sy, -0.5f), -1.0f), yhi);
                bx0 = fminf(bx0, ix); bx1 = fmaxf(bx1, ix);
                by0 = fminf(by0, iy); by1 = fmaxf(by1, iy);
                zmin = fminf(zmin, Z);
            }
        }
        Box bx;
        bx.x0 = (int)floorf(bx0) + 1;  // padded coordinates: pixel -1 is column 0
        bx.y0 = (int)floorf(by0) + 1;
        bx.rw = (int)floorf(bx1) + 3 - bx.x0;  // floor(bx1)+1 (second tap) +1 (padding shift) - x0 + 1
        bx.rh = (int)floorf(by1) + 3 - bx.y0;
        bx.staged = zmin > 1e-6f && bx.rh <= RH_MAX && bx.rw * Q <= ROWQ_MAX;
        return bx;
    };
    auto src_image = [&](int v) {
        return reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.src.p[v]) + (size_t)b * img_bytes);
    };
    Box cur = footprint(0);
    if (cur.staged) {
        lds_dma_copy<NROW, NCOL>(region[0], src_image(0) + ((size_t)cur.y0 * W2 + cur.x0) * Q, W2 * Q, cur.rw * Q, cur.rh, wave, lane);
    }
    __syncthreads();

    for (int v = 0; v < p.V; ++v) {
        Box nxt = cur;
        const bool has_next = v + 1 < p.V;
        if (has_next) {
            nxt = footprint(v + 1);
            if (nxt.staged)  // streams into the other buffer while this view's taps are computed
                lds_dma_copy<NROW, NCOL>(region[(v + 1) & 1], src_image(v + 1) + ((size_t)nxt.y0 * W2 + nxt.x0) * Q, W2 * Q,
                                         nxt.rw * Q, nxt.rh, wave, lane);
        }
        const float* __restrict__ M = p.M + ((size_t)v * p.B + b) * 12;
        const float4* __restrict__ reg = region[v & 1];
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(p.src.p[v]) + (size_t)b * img_bytes), 0, (int)img_bytes,
            0x00020000);
#pragma unroll
        for (int pi = 0; pi < 1; ++pi) {
            const float fy = (float)ys[pi];
            const float ax = fmaf(M[0], fxs, fmaf(M[1], fy, M[2]));
            const float ay = fmaf(M[4], fxs, fmaf(M[5], fy, M[6]));
            const float az = fmaf(M[8], fxs, fmaf(M[9], fy, M[10]));
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const float X = fmaf(ax, dep[i], M[3]), Y = fmaf(ay, dep[i], M[7]), Z = fmaf(az, dep[i], M[11]);
                const float rz = __builtin_amdgcn_rcpf(Z);
                const float ix = fminf(fmaxf(fmaf(X * rz, sx, -0.5f), -1.0f), xhi);
                const float iy = fminf(fmaxf(fmaf(Y * rz, sy, -0.5f), -1.0f), yhi);
                const float xf = floorf(ix), yf = floorf(iy);
                const float wx = ix - xf, wy = iy - yf;
                float4 f00, f10, f01, f11;
                if (cur.staged) {
                    // padded coordinates relative to the staged box, clamped into it
                    const int xi = (int)xf + 1 - cur.x0, yi = (int)yf + 1 - cur.y0;
                    const int xa = min(max(xi, 0), cur.rw - 1), xb = min(max(xi + 1, 0), cur.rw - 1);
                    const int ya = min(max(yi, 0), cur.rh - 1), yb = min(max(yi + 1, 0), cur.rh - 1);
                    const int ra = ya * LDS_ROWQ + q, rb = yb * LDS_ROWQ + q;
                    f00 = reg[ra + xa * Q];
                    f10 = reg[ra + xb * Q];
                    f01 = reg[rb + xa * Q];
                    f11 = reg[rb + xb * Q];
                } else {
                    const unsigned off = rowb + PIX + (unsigned)q * 16 + (unsigned)((int)yf * W2 + (int)xf) * PIX;
                    const u32x4 a0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
                    const u32x4 a1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + PIX, 0, 0);
                    const u32x4 a2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + rowb, 0, 0);
                    const u32x4 a3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + rowb + PIX, 0, 0);
                    f00 = make_float4(__uint_as_float(a0.x), __uint_as_float(a0.y), __uint_as_float(a0.z), __uint_as_float(a0.w));
                    f10 = make_float4(__uint_as_float(a1.x), __uint_as_float(a1.y), __uint_as_float(a1.z), __uint_as_float(a1.w));
                    f01 = make_float4(__uint_as_float(a2.x), __uint_as_float(a2.y), __uint_as_float(a2.z), __uint_as_float(a2.w));
                    f11 = make_float4(__uint_as_float(a3.x), __uint_as_float(a3.y), __uint_as_float(a3.z), __uint_as_float(a3.w));
                }
                const float ux = 1.0f - wx, uy = 1.0f - wy;
                const float w00 = ux * uy, w10 = wx * uy, w01 = ux * wy, w11 = wx * wy;
                float4 acc;  // same accumulation order as the direct kernel (bit-identical results)
                acc.x = fmaf(f11.x, w11, fmaf(f01.x, w01, fmaf(f10.x, w10, fmaf(f00.x, w00, 0.0f))));
                acc.y = fmaf(f11.y, w11, fmaf(f01.y, w01, fmaf(f10.y, w10, fmaf(f00.y, w00, 0.0f))));
                acc.z = fmaf(f11.z, w11, fmaf(f01.z, w01, fmaf(f10.z, w10, fmaf(f00.z, w00, 0.0f))));
                acc.w = fmaf(f11.w, w11, fmaf(f01.w, w01, fmaf(f10.w, w10, fmaf(f00.w, w00, 0.0f))));
                s1[pi][i].x += acc.x; s1[pi][i].y += acc.y; s1[pi][i].z += acc.z; s1[pi][i].w += acc.w;
                s2[pi][i].x = fmaf(acc.x, acc.x, s2[pi][i].x); s2[pi][i].y = fmaf(acc.y, acc.y, s2[pi][i].y);
                s2[pi][i].z = fmaf(acc.z, acc.z, s2[pi][i].z); s2[pi][i].w = fmaf(acc.w, acc.w, s2[pi][i].w);
            }
        }
        __syncthreads();  // (drains the LDS-DMA) buffer (v+1)&1 is complete; buffer v&1 is free for view v+2
        cur = nxt;
    }

    const float inv_nv = 1.0f / (float)(p.V + 1);
#pragma unroll
    for (int pi = 0; pi < 1; ++pi) {
        const int y = y0 + ly0, x = x0 + lx;
        if (y >= h || x >= w) continue;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int d = d0 + i;
            if (d >= D) break;
            const float mx = s1[pi][i].x * inv_nv, my = s1[pi][i].y * inv_nv, mz = s1[pi][i].z * inv_nv,
                        mw = s1[pi][i].w * inv_nv;
            const float4 r = make_float4(fmaf(s2[pi][i].x, inv_nv, -mx * mx), fmaf(s2[pi][i].y, inv_nv, -my * my),
                                         fmaf(s2[pi][i].z, inv_nv, -mz * mz), fmaf(s2[pi][i].w, inv_nv, -mw * mw));
            *reinterpret_cast<float4*>(p.out + ((((size_t)b * D + d) * h + y) * w + x) * C + q * 4) = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Wave-autonomous LDS form: like warp_variance_lds_kernel, but every WAVE owns its own 4x2 key tile and its own
// pair of LDS buffers, so there is no workgroup barrier anywhere: a wave prefetches the footprint of view v+1 with
// LDS-DMA into its second buffer, waits only for the OLDER copy with a counted `s_waitcnt vmcnt(N)` (the LDS-DMA
// of the next view stays in flight) and computes view v from its first buffer.  Eight such waves per CU overlap
// each other's latencies.  Lanes: 8 pixels (4 wide x 2 high) x 8 channel quads; ND planes per task.
template <int ND>
__global__ void __launch_bounds__(256, 2) warp_variance_wave_kernel(WarpParams p) {
    constexpr int C = 32, Q = 8, TXW = 4, TYW = 2;
    constexpr unsigned PIX = 128;
    constexpr int RH = 4, ROWQ = 128;              // staged box: up to 4 rows x 16 pixels, row pitch 128 float4
    constexpr int BUF = RH * ROWQ;                 // float4s per buffer (8 KiB)
    constexpr int NDMA = RH * (ROWQ / 64);         // LDS-DMA instructions per view (always all of them: fixed count)
    extern __shared__ __attribute__((aligned(16))) float4 wlds[];  // [4 waves][2][BUF] = 64 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int q = lane & 7, pp = lane >> 3;
    const int lx = pp & 3, ly = pp >> 2;
    const int h = p.h, w = p.w, D = p.D;
    float4* __restrict__ mybuf = wlds + wave * 2 * BUF;

    // ---- task decode: xcd | d-chunk fastest | tile group (4 x-adjacent wave tiles per workgroup) | batch ----
    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    const int dchunks = (D + ND - 1) / ND;
    const int dc = j % dchunks; j /= dchunks;
    const int grp_in = j % p.tiles_per_xcd;
    const int b = j / p.tiles_per_xcd;
    const int grp = xcd * p.tiles_per_xcd + grp_in;   // group of 4 wave tiles = 16 x 2 pixels
    if (grp >= p.tiles_x * p.tiles_y) return;          // block-uniform
    const int gy = grp / p.tiles_x;
    const int x0 = (grp - gy * p.tiles_x) * (4 * TXW) + wave * TXW, y0 = gy * TYW;
    if (x0 >= w) return;                               // wave-uniform (ragged right edge)
    const int d0 = dc * ND;
    const int dl = min(d0 + ND, D) - 1;

    const float sx = (float)w / (float)(w - 1), sy = (float)h / (float)(h - 1);
    const float xhi = (float)w, yhi = (float)h;
    const int W2 = w + 3;
    const unsigned rowb = (unsigned)W2 * PIX;
    const unsigned img_bytes = (unsigned)(h + 3) * rowb;
    const int xs = min(x0 + lx, w - 1), ysr = min(y0 + ly, h - 1);
    const float fxs = (float)xs, fys = (float)ysr;
    const float* __restrict__ dvals = p.depth + (size_t)b * D;
    float dep[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) dep[i] = dvals[min(d0 + i, D - 1)];
    const float dfirst = dvals[d0], dlast = dvals[dl];
    const float cxs[2] = {(float)x0, (float)min(x0 + TXW - 1, w - 1)};
    const float cys[2] = {(float)y0, (float)min(y0 + TYW - 1, h - 1)};

    float4 s1[ND], s2[ND];
    {
        const float4 k = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.key) + (size_t)b * img_bytes +
                                                          (unsigned)(ysr + 1) * rowb + (unsigned)(xs + 1) * PIX + q * 16);
        const float4 k2 = make_float4(k.x * k.x, k.y * k.y, k.z * k.z, k.w * k.w);
#pragma unroll
        for (int i = 0; i < ND; ++i) { s1[i] = k; s2[i] = k2; }
    }

    struct Box { int x0, y0, rw, rh; bool staged; };
    auto footprint = [&](int v) {
        const float* __restrict__ M = p.M + ((size_t)v * p.B + b) * 12;
        float bx0 = 3e38f, bx1 = -3e38f, by0 = 3e38f, by1 = -3e38f, zmin = 3e38f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float cx = cxs[a & 1], cy = cys[a >> 1];
            const float ax = fmaf(M[0], cx, fmaf(M[1], cy, M[2])), ay = fmaf(M[4], cx, fmaf(M[5], cy, M[6]));
            const float az = fmaf(M[8], cx, fmaf(M[9], cy, M[10]));
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float cd = e ? dlast : dfirst;
                const float X = fmaf(ax, cd, M[3]), Y = fmaf(ay, cd, M[7]), Z = fmaf(az, cd, M[11]);
                const float rz = __builtin_amdgcn_rcpf(Z);
                const float ix = fminf(fmaxf(fmaf(X * rz, sx, -0.5f), -1.0f), xhi);
                const float iy = fminf(fmaxf(fmaf(Y * rz, sy, -0.5f), -1.0f), yhi);
                bx0 = fminf(bx0, ix); bx1 = fmaxf(bx1, ix);
                by0 = fminf(by0, iy); by1 = fmaxf(by1, iy);
                zmin = fminf(zmin, Z);
            }
        }
        Box bx;
        bx.x0 = (int)floorf(bx0) + 1;  // padded coordinates
        bx.y0 = (int)floorf(by0) + 1;
        bx.rw = (int)floorf(bx1) + 3 - bx.x0;
        bx.rh = (int)floorf(by1) + 3 - bx.y0;
        bx.staged = zmin > 1e-6f && bx.rh <= RH && bx.rw * Q <= ROWQ;
        // wave-uniform by construction (computed from wave-uniform inputs); make it so for the compiler too
        bx.x0 = __builtin_amdgcn_readfirstlane(bx.x0);
        bx.y0 = __builtin_amdgcn_readfirstlane(bx.y0);
        bx.rw = __builtin_amdgcn_readfirstlane(bx.rw);
        bx.rh = __builtin_amdgcn_readfirstlane(bx.rh);
        bx.staged = __builtin_amdgcn_readfirstlane((int)bx.staged) != 0;
        return bx;
    };
    // exactly NDMA LDS-DMA instructions per call (rows / columns beyond the box re-read a valid element)
    auto dma = [&](int v, const Box& bx, float4* __restrict__ dst) {
        const float4* __restrict__ g = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.src.p[v]) +
                                                                         (size_t)b * img_bytes) + ((size_t)bx.y0 * W2 + bx.x0) * Q;
        const int rowq = bx.rw * Q;
#pragma unroll
        for (int r = 0; r < RH; ++r) {
            const int row = min(r, bx.rh - 1);
#pragma unroll
            for (int c = 0; c < ROWQ / 64; ++c)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(g + (size_t)row * W2 * Q + min(lane + 64 * c, rowq - 1)),
                    (__attribute__((address_space(3))) void*)(dst + r * ROWQ + c * 64), 16, 0, 0);
        }
    };

    Box cur = footprint(0);
    if (cur.staged) dma(0, cur, mybuf);

    for (int v = 0; v < p.V; ++v) {
        Box nxt = cur;
        bool next_dma = false;
        if (v + 1 < p.V) {
            nxt = footprint(v + 1);
            next_dma = nxt.staged;
            if (next_dma) dma(v + 1, nxt, mybuf + ((v + 1) & 1) * BUF);  // streams in while view v is computed
        }
        // the copy of view v must have landed; the NDMA newer instructions (view v+1) may stay in flight
        if (next_dma) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        const float* __restrict__ M = p.M + ((size_t)v * p.B + b) * 12;
        const float4* __restrict__ reg = mybuf + (v & 1) * BUF;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(p.src.p[v]) + (size_t)b * img_bytes), 0, (int)img_bytes,
            0x00020000);
        const float ax = fmaf(M[0], fxs, fmaf(M[1], fys, M[2]));
        const float ay = fmaf(M[4], fxs, fmaf(M[5], fys, M[6]));
        const float az = fmaf(M[8], fxs, fmaf(M[9], fys, M[10]));
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const float X = fmaf(ax, dep[i], M[3]), Y = fmaf(ay, dep[i], M[7]), Z = fmaf(az, dep[i], M[11]);
            const float rz = __builtin_amdgcn_rcpf(Z);
            const float ix = fminf(fmaxf(fmaf(X * rz, sx, -0.5f), -1.0f), xhi);
            const float iy = fminf(fmaxf(fmaf(Y * rz, sy, -0.5f), -1.0f), yhi);
            const float xf = floorf(ix), yf = floorf(iy);
            const float wx = ix - xf, wy = iy - yf;
            float4 f00, f10, f01, f11;
            if (cur.staged) {
                const int xi = (int)xf + 1 - cur.x0, yi = (int)yf + 1 - cur.y0;
                const int xa = min(max(xi, 0), cur.rw - 1), xb = min(max(xi + 1, 0), cur.rw - 1);
                const int ya = min(max(yi, 0), cur.rh - 1), yb = min(max(yi + 1, 0), cur.rh - 1);
                const int ra = ya * ROWQ + q, rb = yb * ROWQ + q;
                f00 = reg[ra + xa * Q];
                f10 = reg[ra + xb * Q];
                f01 = reg[rb + xa * Q];
                f11 = reg[rb + xb * Q];
            } else {
                const unsigned off = rowb + PIX + (unsigned)q * 16 + (unsigned)((int)yf * W2 + (int)xf) * PIX;
                const u32x4 a0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
                const u32x4 a1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + PIX, 0, 0);
                const u32x4 a2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + rowb, 0, 0);
                const u32x4 a3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + rowb + PIX, 0, 0);
                f00 = make_float4(__uint_as_float(a0.x), __uint_as_float(a0.y), __uint_as_float(a0.z), __uint_as_float(a0.w));
                f10 = make_float4(__uint_as_float(a1.x), __uint_as_float(a1.y), __uint_as_float(a1.z), __uint_as_float(a1.w));
                f01 = make_float4(__uint_as_float(a2.x), __uint_as_float(a2.y), __uint_as_float(a2.z), __uint_as_float(a2.w));
                f11 = make_float4(__uint_as_float(a3.x), __uint_as_float(a3.y), __uint_as_float(a3.z), __uint_as_float(a3.w));
            }
            const float ux = 1.0f - wx, uy = 1.0f - wy;
            const float w00 = ux * uy, w10 = wx * uy, w01 = ux * wy, w11 = wx * wy;
            float4 acc;  // same accumulation order as the direct kernel (bit-identical results)
            acc.x = fmaf(f11.x, w11, fmaf(f01.x, w01, fmaf(f10.x, w10, fmaf(f00.x, w00, 0.0f))));
            acc.y = fmaf(f11.y, w11, fmaf(f01.y, w01, fmaf(f10.y, w10, fmaf(f00.y, w00, 0.0f))));
            acc.z = fmaf(f11.z, w11, fmaf(f01.z, w01, fmaf(f10.z, w10, fmaf(f00.z, w00, 0.0f))));
            acc.w = fmaf(f11.w, w11, fmaf(f01.w, w01, fmaf(f10.w, w10, fmaf(f00.w, w00, 0.0f))));
            s1[i].x += acc.x; s1[i].y += acc.y; s1[i].z += acc.z; s1[i].w += acc.w;
            s2[i].x = fmaf(acc.x, acc.x, s2[i].x); s2[i].y = fmaf(acc.y, acc.y, s2[i].y);
            s2[i].z = fmaf(acc.z, acc.z, s2[i].z); s2[i].w = fmaf(acc.w, acc.w, s2[i].w);
        }
        cur = nxt;
    }

    const float inv_nv = 1.0f / (float)(p.V + 1);
    const int y = y0 + ly, x = x0 + lx;
    if (y < h && x < w) {
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int d = d0 + i;
            if (d >= D) break;
            const float mx = s1[i].x * inv_nv, my = s1[i].y * inv_nv, mz = s1[i].z * inv_nv, mw = s1[i].w * inv_nv;
            const float4 r = make_float4(fmaf(s2[i].x, inv_nv, -mx * mx), fmaf(s2[i].y, inv_nv, -my * my),
                                         fmaf(s2[i].z, inv_nv, -mz * mz), fmaf(s2[i].w, inv_nv, -mw * mw));
            *reinterpret_cast<float4*>(p.out + ((((size_t)b * D + d) * h + y) * w + x) * C + q * 4) = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Experimental (MVD_K3_CFG=q8): C = 32, channel-last output, folded grid arithmetic, ONE QUAD PER KEY PIXEL with 8
// channels per lane and 2 planes per workgroup.  The per-(pixel, plane, view) overhead of the direct kernel (position
// arithmetic, DPP broadcasts, re-gather pattern, addresses) is amortised over twice the FMAs per lane; lanes q and q^2
// of a quad both locate plane q & 1.  Bit-identical to the direct kernel.
__device__ __forceinline__ void gather_cell_q8(u32x4 (&f)[4], u32x4 (&g)[4], __amdgpu_buffer_rsrc_t rsrc, unsigned o, unsigned rowb) {
    constexpr unsigned PIX = 128;
    f[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);
    g[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + 16, 0, 0);
    f[1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + PIX, 0, 0);
    g[1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + PIX + 16, 0, 0);
    f[2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + rowb, 0, 0);
    g[2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + rowb + 16, 0, 0);
    f[3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + rowb + PIX, 0, 0);
    g[3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + rowb + PIX + 16, 0, 0);
}

template <bool ANY1>
__device__ __forceinline__ void step_q8(float4 (&s1)[2][2], float4 (&s2)[2][2], const float (&w0)[4], const float (&w1)[4],
                                        unsigned o0, unsigned o1, __amdgpu_buffer_rsrc_t rsrc, unsigned rowb) {
    u32x4 f0[4], g0[4], f1[4], g1[4];
    gather_cell_q8(f0, g0, rsrc, o0, rowb);
    if constexpr (ANY1) gather_cell_q8(f1, g1, rsrc, o1, rowb);
    accumulate_cell(s1[0][0], s2[0][0], w0, f0);
    accumulate_cell(s1[0][1], s2[0][1], w0, g0);
    accumulate_cell(s1[1][0], s2[1][0], w1, ANY1 ? f1 : f0);
    accumulate_cell(s1[1][1], s2[1][1], w1, ANY1 ? g1 : g0);
}

template <int MINW, int CPB>
__global__ void __launch_bounds__(256, MINW) warp_variance_q8_kernel(WarpParams p) {
    constexpr unsigned PIX = 128;
    const int tid = threadIdx.x;
    const int q = tid & 3;    // channels 8q .. 8q+7; locates plane d0 + (q & 1)
    const int px = tid >> 2;  // 0..63: 32 columns x 2 rows
    const int h = p.h, w = p.w, D = p.D;

    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    const int dchunks = (D + 2 * CPB - 1) / (2 * CPB);  // CPB chunks of 2 planes per workgroup: one index decode and key fetch for all
    const int dc = j % dchunks; j /= dchunks;
    const int tile_in = j % p.tiles_per_xcd;
    const int b = j / p.tiles_per_xcd;
    const int tile = xcd * p.tiles_per_xcd + tile_in;
    if (tile >= p.tiles_x * p.tiles_y) return;  // block-uniform
    const int ty = tile / p.tiles_x;
    const int x = (tile - ty * p.tiles_x) * 32 + (px & 31);
    const int y = ty * 2 + (px >> 5);
    const bool active = x < w && y < h;
    const int xc = min(x, w - 1), yc = min(y, h - 1);

    const float sx = (float)w / (float)(w - 1), sy = (float)h / (float)(h - 1);
    const float fx = (float)xc, fy = (float)yc;
    const float xhi = (float)w, yhi = (float)h;
    const int W2 = w + 3;
    const float W2f = (float)W2;
    const unsigned rowb = (unsigned)W2 * PIX;
    const unsigned img_bytes = (unsigned)(h + 3) * rowb;
    const unsigned org = rowb + PIX + (unsigned)q * 32;

    float4 k0, k1;
    {
        const float4* kp = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.key) + (size_t)b * img_bytes + org +
                                                           (unsigned)yc * rowb + (unsigned)xc * PIX);
        k0 = kp[0]; k1 = kp[1];
    }
    const float inv_nv = 1.0f / (float)(p.V + 1);  // mvsnet.py:135, V there counts the key view
#pragma unroll 1
    for (int cc = 0; cc < CPB; ++cc) {
    const int d0 = (dc * CPB + cc) * 2;
    if (d0 >= D) break;  // block-uniform
    float4 s1[2][2], s2[2][2];  // [plane][channel half]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        s1[i][0] = k0; s1[i][1] = k1;
        s2[i][0] = make_float4(k0.x * k0.x, k0.y * k0.y, k0.z * k0.z, k0.w * k0.w);
        s2[i][1] = make_float4(k1.x * k1.x, k1.y * k1.y, k1.z * k1.z, k1.w * k1.w);
    }
    const float mydep = p.depth[(size_t)b * D + min(d0 + (q & 1), D - 1)];

    float Mn[12];
    const char* srcn;
    auto fetch_view = [&](int v) {
        const float* __restrict__ Mv = p.M + ((size_t)v * p.B + b) * 12;  // wave-uniform: scalar loads
#pragma unroll
        for (int k = 0; k < 12; ++k) Mn[k] = Mv[k];
        srcn = reinterpret_cast<const char*>(p.src.p[v]);
    };
    fetch_view(0);
    for (int v = 0; v < p.V; ++v) {
        float M[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) M[k] = Mn[k];
        const char* srcv = srcn;
        fetch_view(min(v + 1, p.V - 1));
        const float ax = fmaf(M[0], fx, fmaf(M[1], fy, M[2]));
        const float ay = fmaf(M[4], fx, fmaf(M[5], fy, M[6]));
        const float az = fmaf(M[8], fx, fmaf(M[9], fy, M[10]));
        const float X = fmaf(ax, mydep, M[3]), Y = fmaf(ay, mydep, M[7]), Z = fmaf(az, mydep, M[11]);
        const float rz = __builtin_amdgcn_rcpf(Z);
        float ix = fmaf(X * rz, sx, -0.5f), iy = fmaf(Y * rz, sy, -0.5f);
        ix = __builtin_amdgcn_fmed3f(ix, -1.0f, xhi);
        iy = __builtin_amdgcn_fmed3f(iy, -1.0f, yhi);
        const float xf = floorf(ix), yf = floorf(iy);
        const float mwx = ix - xf, mwy = iy - yf;
        const unsigned mpo = (unsigned)(int)fmaf(yf, W2f, xf) * PIX;
        const float mux = 1.0f - mwx, muy = 1.0f - mwy;
        const float m00 = mux * muy, m10 = mwx * muy, m01 = mux * mwy, m11 = mwx * mwy;
        float wt[2][4];
        unsigned off[2];
#define MVD_QB(V, I) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(V), (I) * 0x55, 0xf, 0xf, true))
#define MVD_QUAD_BCAST(I)                                                                                          \
    wt[I][0] = MVD_QB(m00, I); wt[I][1] = MVD_QB(m10, I); wt[I][2] = MVD_QB(m01, I); wt[I][3] = MVD_QB(m11, I);     \
    off[I] = org + (unsigned)__builtin_amdgcn_mov_dpp((int)mpo, (I) * 0x55, 0xf, 0xf, true);  /* quad_perm:[I,I,I,I] */
        MVD_QUAD_BCAST(0) MVD_QUAD_BCAST(1)
#undef MVD_QUAD_BCAST
#undef MVD_QB
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(srcv + (size_t)b * img_bytes), 0, (int)img_bytes, 0x00020000);
        if (__builtin_amdgcn_ballot_w64(off[1] != off[0]) != 0) step_q8<true>(s1, s2, wt[0], wt[1], off[0], off[1], rsrc, rowb);
        else step_q8<false>(s1, s2, wt[0], wt[1], off[0], off[1], rsrc, rowb);
    }

    if (active) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int d = d0 + i;
        if (d >= D) break;  // block-uniform
        float4* op = reinterpret_cast<float4*>(p.out + ((((size_t)b * D + d) * h + y) * w + x) * 32 + q * 8);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const float mx = s1[i][hf].x * inv_nv, my = s1[i][hf].y * inv_nv, mz = s1[i][hf].z * inv_nv, mw = s1[i][hf].w * inv_nv;
            op[hf] = make_float4(fmaf(s2[i][hf].x, inv_nv, -mx * mx), fmaf(s2[i][hf].y, inv_nv, -my * my),
                                 fmaf(s2[i][hf].z, inv_nv, -mz * mz), fmaf(s2[i][hf].w, inv_nv, -mw * mw));
        }
    }
    }
    }
}

int launch_warp_q8(const WarpParams& p0, hipStream_t st, int minw) {
    WarpParams p = p0;
    p.tiles_x = (p.w + 31) / 32;
    p.tiles_y = (p.h + 1) / 2;
    const long long tiles = (long long)p.tiles_x * p.tiles_y;
    p.tiles_per_xcd = (int)((tiles + 7) / 8);
    // (several 2-plane chunks per workgroup, to amortise the index decode and key fetch, measured 1.5 ms: CPB stays 1)
    const long long nblk = 8LL * p.tiles_per_xcd * ((p.D + 1) / 2) * p.B;
    if (nblk > 0x7fffffffLL) {
        set_error("warp_variance: %lld workgroups exceed the grid limit", nblk);
        return MVD_ERR_INVALID_ARG;
    }
    timing_begin(st);
    const dim3 grid((unsigned)nblk);
    if (minw == 3) hipLaunchKernelGGL((warp_variance_q8_kernel<3, 1>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((warp_variance_q8_kernel<4, 1>), grid, dim3(256), 0, st, p);
    timing_end(st);
    return launch_status("warp_variance_q8");
}

int launch_warp_wave(const WarpParams& p0, hipStream_t st, int nd) {
    WarpParams p = p0;
    p.tiles_x = (p.w + 15) / 16;  // groups of 4 wave tiles (16 x 2 pixels)
    p.tiles_y = (p.h + 1) / 2;
    const long long groups = (long long)p.tiles_x * p.tiles_y;
    p.tiles_per_xcd = (int)((groups + 7) / 8);
    const long long nblk = 8LL * p.tiles_per_xcd * ((p.D + nd - 1) / nd) * p.B;
    if (nblk > 0x7fffffffLL) {
        set_error("warp_variance: %lld workgroups exceed the grid limit", nblk);
        return MVD_ERR_INVALID_ARG;
    }
    const size_t lds = (size_t)4 * 2 * 4 * 128 * sizeof(float4);
    timing_begin(st);
    switch (nd) {
#define MVD_W(ND)                                                                                                  \
    case ND:                                                                                                       \
        (void)hipFuncSetAttribute((const void*)warp_variance_wave_kernel<ND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(warp_variance_wave_kernel<ND>, dim3((unsigned)nblk), dim3(256), lds, st, p);            \
        break;
        MVD_W(4) MVD_W(8)
#undef MVD_W
        default:
            set_error("warp_variance: MVD_K3_CFG wave,%d is not a compiled variant", nd);
            return MVD_ERR_INVALID_ARG;
    }
    timing_end(st);
    return launch_status("warp_variance_wave");
}

int launch_warp_lds(const WarpParams& p0, hipStream_t st, int nd) {
    WarpParams p = p0;
    p.tiles_x = (p.w + 7) / 8;
    p.tiles_y = (p.h + 3) / 4;
    const long long tiles = (long long)p.tiles_x * p.tiles_y;
    p.tiles_per_xcd = (int)((tiles + 7) / 8);
    const long long nblk = 8LL * p.tiles_per_xcd * ((p.D + nd - 1) / nd) * p.B;
    if (nblk > 0x7fffffffLL) {
        set_error("warp_variance: %lld workgroups exceed the grid limit", nblk);
        return MVD_ERR_INVALID_ARG;
    }
    timing_begin(st);
    switch (nd) {
        case 2: hipLaunchKernelGGL(warp_variance_lds_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, st, p); break;
        case 4: hipLaunchKernelGGL(warp_variance_lds_kernel<4>, dim3((unsigned)nblk), dim3(256), 0, st, p); break;
        case 8: hipLaunchKernelGGL(warp_variance_lds_kernel<8>, dim3((unsigned)nblk), dim3(256), 0, st, p); break;
        default:
            set_error("warp_variance: MVD_K3_CFG lds,%d is not a compiled variant", nd);
            return MVD_ERR_INVALID_ARG;
    }
    timing_end(st);
    return launch_status("warp_variance_lds");
}

}  // namespace mvd
