// K3 — fronto-parallel homography warp of V source feature maps into the key frustum at D depth
// hypotheses + variance aggregation with the key features, in one pass (Path B).
// Replaces homo_warp (rmvd/models/blocks/utils.py:222-268) and the sum / sum-of-squares / variance
// arithmetic of MVSNet.forward (rmvd/models/mvsnet.py:124-136).
//
// HBM-bound by construction: the only large tensor is the variance volume, written exactly once
// (algorithmic bytes 4*((V+1)*C*h*w + C*D*h*w) per batch element); the per-view warped volumes of the
// reference are never materialised.  Feature maps are first repacked channel-last with a zero border
// ((h+3,w+3,C), 128 B per pixel at C=32): one bilinear tap of one pixel is a single full cache line shared
// by C/4 lanes, and zero padding costs no in-bounds logic (the clamped sample position lands on zero taps).
//
// Thread mapping: C/4 lanes per output pixel (each lane owns 4 channels = one 16-B load per tap),
// 256/(C/4) consecutive x pixels per workgroup, one (b, d, y) row segment per workgroup.
#include "mvd_common.h"
#include <type_traits>
#include "warp_variance_common.h"
// knock-out builds (parts of the kernel removed to time the rest; they compute WRONG results) exist only in the
// experiments library
#if !defined(MVD_EXPERIMENTS) || !defined(MVD_K3_EXPERIMENT)
#undef MVD_K3_EXPERIMENT
#define MVD_K3_EXPERIMENT 0
#endif

namespace mvd {


// M[v][b] = (src_proj[v][b] @ key_proj_inv[b])[:3,:4] as an fmaf chain over k (what a K=4 sgemm does):
// computed once per call by a one-block prologue kernel; the main kernel reads the 12 floats through the
// scalar cache instead of redoing a uniform 4x4 product in every lane.
__global__ void compose_transforms_kernel(ViewPtrs proj, const float* __restrict__ key_proj_inv, int B, int V,
                                          float* __restrict__ M) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= V * B * 12) return;
    const int j = e % 4, i = (e / 4) % 3, b = (e / 12) % B, v = e / (12 * B);
    const float* P = proj.p[v] + b * 16;
    const float* Q = key_proj_inv + b * 16;
    float acc = P[i * 4 + 0] * Q[0 * 4 + j];
    acc = fmaf(P[i * 4 + 1], Q[1 * 4 + j], acc);
    acc = fmaf(P[i * 4 + 2], Q[2 * 4 + j], acc);
    acc = fmaf(P[i * 4 + 3], Q[3 * 4 + j], acc);
    M[e] = acc;
}

// (N, C, h, w) -> channel-last with a zero border: (N, h+3, w+3, C), pixel (y, x) at padded (y+1, x+1).
// Rows/cols -1, w and w+1 (h, h+1) stay zero (the buffer is cleared first), so a bilinear sample whose
// position is clamped to [-1, w] x [-1, h] needs no in-bounds logic: out-of-image taps read zeros.
__global__ void __launch_bounds__(256) repack_padded_kernel(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                            int h, int w) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const long long hw = (long long)h * w;
    const long long p0 = (long long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k;
        const long long pix = p0 + tx;
        if (c < C && pix < hw) tile[ty + 8 * k][tx] = src[((long long)n * C + c) * hw + pix];
    }
    __syncthreads();
    const int W2 = w + 3;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long pix = p0 + ty + 8 * k;
        const int c = c0 + tx;
        if (c < C && pix < hw) {
            const int y = (int)(pix / w), x = (int)(pix - (long long)y * w);
            dst[(((long long)n * (h + 3) + y + 1) * W2 + x + 1) * C + c] = tile[tx][ty + 8 * k];
        }
    }
}


// The 4 planes of a workgroup for one source view when the WAVE's re-gather pattern is MASK (bit i-1: some lane's
// 2x2 cell differs between plane i-1 and plane i).  Planes whose bit is clear reuse the previous plane's registers
// outright — no per-lane select, no exec masking: the texture path charges a gather instruction the same whether 1
// or 64 lanes are active, so re-gathering for the whole wave costs nothing extra, and a pattern known at compile time
// lets every load be issued up front and waited for with exact counts.
template <int MASK>
__device__ __forceinline__ void gather_blend_4planes(float4 (&s1)[4], float4 (&s2)[4], const float (&wt)[4][4],
                                                     const unsigned (&off)[4], __amdgpu_buffer_rsrc_t rsrc, unsigned rowb,
                                                     unsigned pix) {
    u32x4 f[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i == 0 || ((MASK >> (i - 1)) & 1)) {
#if MVD_K3_EXPERIMENT == 1 || MVD_K3_EXPERIMENT == 2
            f[i][0] = u32x4{off[i], off[i] + 1, off[i] + 2, off[i] + 3};
            f[i][1] = u32x4{off[i] + pix, off[i] + 5, off[i] + 6, off[i] + 7};
            f[i][2] = u32x4{off[i] + rowb, off[i] + 9, off[i] + 10, off[i] + 11};
            f[i][3] = u32x4{off[i] + rowb + pix, off[i] + 13, off[i] + 14, off[i] + 15};
#else
            f[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[i], 0, 0);
            f[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[i] + pix, 0, 0);
            f[i][2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[i] + rowb, 0, 0);
            f[i][3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[i] + rowb + pix, 0, 0);
#endif
        }
    }
    int src = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i > 0 && ((MASK >> (i - 1)) & 1)) src = i;
        accumulate_cell(s1[i], s2[i], wt[i], f[src]);
    }
}

// The same for four waves per SIMD (128 VGPRs): the bilinear weights stay in the locating lane until a plane's
// arithmetic needs them (4 DPP broadcasts right there instead of 16 registers held through the gathers), and at most
// three cells are in flight — when all four planes re-gather, plane 3's loads are issued after plane 0's arithmetic
// into the registers it frees.
template <int MASK, int I>
__device__ __forceinline__ void blend_plane_late(float4 (&s1)[4], float4 (&s2)[4], float m00, float m10, float m01, float m11,
                                                 const u32x4 (&f)[4]) {
    float w[4];
    w[0] = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m00), I * 0x55, 0xf, 0xf, true));  // quad_perm:[I,I,I,I]
    w[1] = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m10), I * 0x55, 0xf, 0xf, true));
    w[2] = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m01), I * 0x55, 0xf, 0xf, true));
    w[3] = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m11), I * 0x55, 0xf, 0xf, true));
    accumulate_cell(s1[I], s2[I], w, f);
}

__device__ __forceinline__ void gather_cell(u32x4 (&f)[4], __amdgpu_buffer_rsrc_t rsrc, unsigned o, unsigned rowb, unsigned pix) {
    f[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);
    f[1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + pix, 0, 0);
    f[2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + rowb, 0, 0);
    f[3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + rowb + pix, 0, 0);
}

template <int MASK>
__device__ __forceinline__ void gather_blend_4planes_late(float4 (&s1)[4], float4 (&s2)[4], float m00, float m10, float m01,
                                                          float m11, const unsigned (&off)[4], __amdgpu_buffer_rsrc_t rsrc,
                                                          unsigned rowb, unsigned pix) {
    constexpr bool G1 = MASK & 1, G2 = (MASK >> 1) & 1, G3 = (MASK >> 2) & 1;
    u32x4 a[4], b[4], c[4];  // three register sets
    gather_cell(a, rsrc, off[0], rowb, pix);
    if constexpr (MASK == 7) {
        gather_cell(b, rsrc, off[1], rowb, pix);
        gather_cell(c, rsrc, off[2], rowb, pix);
        blend_plane_late<MASK, 0>(s1, s2, m00, m10, m01, m11, a);
        gather_cell(a, rsrc, off[3], rowb, pix);  // into the set plane 0 just released
        blend_plane_late<MASK, 1>(s1, s2, m00, m10, m01, m11, b);
        blend_plane_late<MASK, 2>(s1, s2, m00, m10, m01, m11, c);
        blend_plane_late<MASK, 3>(s1, s2, m00, m10, m01, m11, a);
    } else {
        // at most two of planes 1..3 re-gather: sets b and c take them in order
        if constexpr (G1) gather_cell(b, rsrc, off[1], rowb, pix);
        if constexpr (G2) gather_cell(G1 ? c : b, rsrc, off[2], rowb, pix);
        if constexpr (G3) gather_cell((G1 || G2) ? c : b, rsrc, off[3], rowb, pix);
        blend_plane_late<MASK, 0>(s1, s2, m00, m10, m01, m11, a);
        const u32x4 (&p1)[4] = G1 ? b : a;
        blend_plane_late<MASK, 1>(s1, s2, m00, m10, m01, m11, p1);
        const u32x4 (&p2)[4] = G2 ? (G1 ? c : b) : p1;
        blend_plane_late<MASK, 2>(s1, s2, m00, m10, m01, m11, p2);
        const u32x4 (&p3)[4] = G3 ? ((G1 || G2) ? c : b) : p2;
        blend_plane_late<MASK, 3>(s1, s2, m00, m10, m01, m11, p3);
    }
}

// Work decomposition (the part that decides where the tap gathers are served from):
//   * a workgroup owns one row segment of PPB key pixels and DPB consecutive depth planes; for each view
//     it computes the DPB sample positions, issues all 4*DPB gathers back to back (buffer loads: SGPR
//     descriptor + one 32-bit offset per sample, the 4 taps at constant strides from it) and only then
//     accumulates, so the gathers of a view overlap each other and consecutive planes touch (nearly) the
//     same source lines;
//   * the 1-D grid is decoded XCD-first (blocks b and b+8 share an XCD, MI355X_MICROARCH.md): each XCD
//     owns a contiguous band of key rows for ALL planes, so its private 4 MiB L2 only ever sees the
//     matching band of each source image (1/8 of 7 MB per view) instead of whole images per plane —
//     with a plane-major grid every XCD streamed all V source images per plane and the gathers were
//     served by the Infinity Cache (measured 2.8 ms at the headline shape, profiles/r01_*).
// Placement only affects speed; results do not depend on it.
template <int LPP, bool WARP_ONLY, int DPB, int MINW, int REUSE = 0, bool EXACT = false>
__global__ void __launch_bounds__(256, MINW) warp_variance_kernel(WarpParams p) {
    constexpr int PPB = 256 / LPP;  // pixels per block
    constexpr int C = LPP * 4;
    constexpr unsigned PIX = LPP * 16;  // bytes per pixel
    __shared__ float stage[C * (PPB + 1)];

    const int tid = threadIdx.x;
    const int q = tid % LPP;   // channel quad
    const int px = tid / LPP;  // pixel within the block
    const int h = p.h, w = p.w, D = p.D;

    // ---- decode the block index: xcd | (d-chunk fastest, then tile within the XCD's band, then batch) ----
    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    const int dchunks = (D + DPB - 1) / DPB;
    const int dc = j % dchunks; j /= dchunks;
    const int tile_in = j % p.tiles_per_xcd;
    const int b = j / p.tiles_per_xcd;
    const int tile = xcd * p.tiles_per_xcd + tile_in;
    if (tile >= p.tiles_x * h) return;  // block-uniform
    const int y = tile / p.tiles_x;
    const int x0 = (tile - y * p.tiles_x) * PPB;
    const int x = x0 + px;
    const int d0 = dc * DPB;
    const bool active = x < w;
    const int xc = active ? x : w - 1;

    // sample index = (X/Z) * w/(w-1) - 0.5: homo_warp's normalisation /((W-1)/2) - 1 followed by
    // grid_sample's ((g+1)*W-1)/2 (utils.py:256-264), folded, with 1/Z from v_rcp_f32 (1 ulp).  Path B has
    // no in-bounds mask, so the few-ulp difference moves a sample by < 1e-4 px and the output continuously.
    const float sx = (float)w / (float)(w - 1), sy = (float)h / (float)(h - 1);
    const float fx = (float)xc, fy = (float)y;
    const float xhi = (float)w, yhi = (float)h;
    const float half_w = (float)(w - 1) / 2.0f, half_h = (float)(h - 1) / 2.0f;
    const int W2 = w + 3;
    const float W2f = (float)W2;
    const unsigned rowb = (unsigned)W2 * PIX;                    // bytes per padded row
    const unsigned img_bytes = (unsigned)(h + 3) * rowb;         // bytes per padded image
    const unsigned org = rowb + PIX + (unsigned)q * 16;          // padded (1,1) + this lane's channel quad

    float4 s1[DPB], s2[DPB];
    if constexpr (!WARP_ONLY) {
        const float4 k = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.key) + (size_t)b * img_bytes +
                                                          org + (unsigned)y * rowb + (unsigned)xc * PIX);
        const float4 k2 = make_float4(k.x * k.x, k.y * k.y, k.z * k.z, k.w * k.w);
#pragma unroll
        for (int i = 0; i < DPB; ++i) { s1[i] = k; s2[i] = k2; }
    } else {
#pragma unroll
        for (int i = 0; i < DPB; ++i) { s1[i] = make_float4(0, 0, 0, 0); s2[i] = s1[i]; }
    }
    const float* __restrict__ dvals = p.depth + (size_t)b * D;
    float dep[DPB];
#pragma unroll
    for (int i = 0; i < DPB; ++i) dep[i] = dvals[min(d0 + i, D - 1)];
    const float mydep = dvals[min(d0 + (q & 3), D - 1)];  // the plane this lane locates for its quad (DPB == 4)

    // the next view's transform and base pointer are fetched (scalar loads) while the current view is processed
    float Mn[12];
    const char* srcn;
    auto fetch_view = [&](int v) {
        const float* __restrict__ Mv = p.M + ((size_t)v * p.B + b) * 12;  // wave-uniform: scalar loads
#pragma unroll
        for (int k = 0; k < 12; ++k) Mn[k] = Mv[k];
        srcn = reinterpret_cast<const char*>(p.src.p[v]);
    };
    fetch_view(0);
#if MVD_K3_EXPERIMENT == 3
    const int nviews = p.V > 100 ? p.V : 0;
#else
    const int nviews = p.V;
#endif
    for (int v = 0; v < nviews; ++v) {
        float M[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) M[k] = Mn[k];
        const char* srcv = srcn;
        fetch_view(min(v + 1, p.V - 1));
        // (X,Y,Z)(d) = R (x,y,1)^T d + T  (utils.py:246-250)
        const float ax = fmaf(M[0], fx, fmaf(M[1], fy, M[2]));
        const float ay = fmaf(M[4], fx, fmaf(M[5], fy, M[6]));
        const float az = fmaf(M[8], fx, fmaf(M[9], fy, M[10]));
        const float tx = M[3], ty = M[7], tz = M[11];
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(srcv + (size_t)b * img_bytes), 0, (int)img_bytes,
            0x00020000);
        unsigned off[DPB];
        float wt[DPB][4];  // bilinear weights of the taps nw, ne, sw, se
        bool moved[DPB];   // REUSE: the 2x2 source cell of plane i differs from plane i-1's
        // position of one plane: clamp into the zero border (a sample outside the image lands on zero taps; NaN from
        // Z == 0 clamps to -1), split into cell and fraction
        auto locate = [&](float depth, float& fwx, float& fwy, unsigned& pixoff) {
            float ix, iy;
            if constexpr (EXACT) {  // a template parameter: as a run-time branch its operands stay live through the loop (16+ VGPRs)
                // the reference's own chain, one rounding per step: R @ (x*d, y*d, d) + T (utils.py:246-250), perspective
                // divide, /((W-1)/2) - 1 (:256-257), grid_sample's ((g+1)*W-1)/2
                const float gx = fx * depth, gy = fy * depth;
                const float X = ((M[0] * gx + M[1] * gy) + M[2] * depth) + tx;
                const float Y = ((M[4] * gx + M[5] * gy) + M[6] * depth) + ty;
                const float Z = ((M[8] * gx + M[9] * gy) + M[10] * depth) + tz;
                ix = unnormalize_coord((X / Z) / half_w - 1.0f, xhi);
                iy = unnormalize_coord((Y / Z) / half_h - 1.0f, yhi);
            } else {
                const float X = fmaf(ax, depth, tx), Y = fmaf(ay, depth, ty), Z = fmaf(az, depth, tz);
                const float rz = __builtin_amdgcn_rcpf(Z);
                ix = fmaf(X * rz, sx, -0.5f);
                iy = fmaf(Y * rz, sy, -0.5f);
            }
            // v_med3_f32: one instruction; with a NaN operand it returns the minimum of the others, i.e. -1 like
            // fminf(fmaxf(NaN, -1), hi)
            ix = __builtin_amdgcn_fmed3f(ix, -1.0f, xhi);
            iy = __builtin_amdgcn_fmed3f(iy, -1.0f, yhi);
            const float xf = floorf(ix), yf = floorf(iy);
            fwx = ix - xf;
            fwy = iy - yf;
            // (yf+1, xf+1) in the padded image once `org` is added.  With 32 channels the padded map has < 2^24 pixels
            // (checked on the host), so yf*W2 + xf is exact in fp32 and replaces a quarter-rate integer multiply.
            if constexpr (LPP == 8) pixoff = (unsigned)(int)fmaf(yf, W2f, xf) * PIX;
            else pixoff = (unsigned)((int)yf * W2 + (int)xf) * PIX;
        };
        if constexpr (DPB == 4 && LPP % 4 == 0) {
            // the LPP lanes of a pixel would each repeat this arithmetic for all 4 planes; instead lane (q & 3) of every
            // quad does plane (q & 3) and the quad exchanges the three results with DPP quad_perm broadcasts
            float mwx, mwy;
            unsigned mpo;
            locate(mydep, mwx, mwy, mpo);
            const float mux = 1.0f - mwx, muy = 1.0f - mwy;
            const float m00 = mux * muy, m10 = mwx * muy, m01 = mux * mwy, m11 = mwx * mwy;
            if constexpr (REUSE == 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) off[i] = 0;
                off[0] = org + (unsigned)__builtin_amdgcn_mov_dpp((int)mpo, 0 * 0x55, 0xf, 0xf, true);
                off[1] = org + (unsigned)__builtin_amdgcn_mov_dpp((int)mpo, 1 * 0x55, 0xf, 0xf, true);
                off[2] = org + (unsigned)__builtin_amdgcn_mov_dpp((int)mpo, 2 * 0x55, 0xf, 0xf, true);
                off[3] = org + (unsigned)__builtin_amdgcn_mov_dpp((int)mpo, 3 * 0x55, 0xf, 0xf, true);
                const unsigned mask = (__builtin_amdgcn_ballot_w64(off[1] != off[0]) != 0 ? 1u : 0u) |
                                      (__builtin_amdgcn_ballot_w64(off[2] != off[1]) != 0 ? 2u : 0u) |
                                      (__builtin_amdgcn_ballot_w64(off[3] != off[2]) != 0 ? 4u : 0u);
                switch (mask) {
                    case 0: gather_blend_4planes_late<0>(s1, s2, m00, m10, m01, m11, off, rsrc, rowb, PIX); break;
                    case 1: gather_blend_4planes_late<1>(s1, s2, m00, m10, m01, m11, off, rsrc, rowb, PIX); break;
                    case 2: gather_blend_4planes_late<2>(s1, s2, m00, m10, m01, m11, off, rsrc, rowb, PIX); break;
                    case 3: gather_blend_4planes_late<3>(s1, s2, m00, m10, m01, m11, off, rsrc, rowb, PIX); break;
                    case 4: gather_blend_4planes_late<4>(s1, s2, m00, m10, m01, m11, off, rsrc, rowb, PIX); break;
                    case 5: gather_blend_4planes_late<5>(s1, s2, m00, m10, m01, m11, off, rsrc, rowb, PIX); break;
                    case 6: gather_blend_4planes_late<6>(s1, s2, m00, m10, m01, m11, off, rsrc, rowb, PIX); break;
                    default: gather_blend_4planes_late<7>(s1, s2, m00, m10, m01, m11, off, rsrc, rowb, PIX); break;
                }
                continue;
            }
#define MVD_QB(V, I) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(V), (I) * 0x55, 0xf, 0xf, true))
#define MVD_QUAD_BCAST(I)                                                                                          \
    wt[I][0] = MVD_QB(m00, I); wt[I][1] = MVD_QB(m10, I); wt[I][2] = MVD_QB(m01, I); wt[I][3] = MVD_QB(m11, I);     \
    off[I] = org + (unsigned)__builtin_amdgcn_mov_dpp((int)mpo, (I) * 0x55, 0xf, 0xf, true);  /* quad_perm:[I,I,I,I] */
            MVD_QUAD_BCAST(0) MVD_QUAD_BCAST(1) MVD_QUAD_BCAST(2) MVD_QUAD_BCAST(3)
#undef MVD_QUAD_BCAST
#undef MVD_QB
#pragma unroll
            for (int i = 0; i < 4; ++i) moved[i] = i == 0 || off[i] != off[i - 1];
        } else {
#pragma unroll
            for (int i = 0; i < DPB; ++i) {
                unsigned po;
                float fwx, fwy;
                locate(dep[i], fwx, fwy, po);
                const float ux = 1.0f - fwx, uy = 1.0f - fwy;
                wt[i][0] = ux * uy; wt[i][1] = fwx * uy; wt[i][2] = ux * fwy; wt[i][3] = fwx * fwy;
                off[i] = org + po;
                moved[i] = i == 0 || off[i] != off[i - 1];
            }
        }
        if constexpr (REUSE == 2 && DPB == 4) {
            // wave-uniform re-gather pattern (see gather_blend_4planes)
            const unsigned mask = (__builtin_amdgcn_ballot_w64(moved[1]) != 0 ? 1u : 0u) |
                                  (__builtin_amdgcn_ballot_w64(moved[2]) != 0 ? 2u : 0u) |
                                  (__builtin_amdgcn_ballot_w64(moved[3]) != 0 ? 4u : 0u);
            switch (mask) {
                case 0: gather_blend_4planes<0>(s1, s2, wt, off, rsrc, rowb, PIX); break;
                case 1: gather_blend_4planes<1>(s1, s2, wt, off, rsrc, rowb, PIX); break;
                case 2: gather_blend_4planes<2>(s1, s2, wt, off, rsrc, rowb, PIX); break;
                case 3: gather_blend_4planes<3>(s1, s2, wt, off, rsrc, rowb, PIX); break;
                case 4: gather_blend_4planes<4>(s1, s2, wt, off, rsrc, rowb, PIX); break;
                case 5: gather_blend_4planes<5>(s1, s2, wt, off, rsrc, rowb, PIX); break;
                case 6: gather_blend_4planes<6>(s1, s2, wt, off, rsrc, rowb, PIX); break;
                default: gather_blend_4planes<7>(s1, s2, wt, off, rsrc, rowb, PIX); break;
            }
            continue;
        }
        u32x4 f[DPB][4];
        if constexpr (REUSE == 1) {
            // Sweep coherence: from one plane to the next a sample moves a fraction of a pixel, so its 2x2 cell is
            // usually the previous plane's.  Only lanes whose cell moved gather again (exec-masked loads, all issued
            // before the first use); the others take the previous plane's registers.
#pragma unroll
            for (int i = 0; i < DPB; ++i) {
                if (moved[i]) {
                    f[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[i], 0, 0);
                    f[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[i] + PIX, 0, 0);
                    f[i][2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[i] + rowb, 0, 0);
                    f[i][3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[i] + rowb + PIX, 0, 0);
                }
            }
#pragma unroll
            for (int i = 1; i < DPB; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f[i][k].x = moved[i] ? f[i][k].x : f[i - 1][k].x;
                    f[i][k].y = moved[i] ? f[i][k].y : f[i - 1][k].y;
                    f[i][k].z = moved[i] ? f[i][k].z : f[i - 1][k].z;
                    f[i][k].w = moved[i] ? f[i][k].w : f[i - 1][k].w;
                }
        } else {
#pragma unroll
            for (int i = 0; i < DPB; ++i) {
                f[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[i], 0, 0);
                f[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[i] + PIX, 0, 0);
                f[i][2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[i] + rowb, 0, 0);
                f[i][3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[i] + rowb + PIX, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < DPB; ++i) {
            float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc.x = fmaf(__uint_as_float(f[i][k].x), wt[i][k], acc.x);
                acc.y = fmaf(__uint_as_float(f[i][k].y), wt[i][k], acc.y);
                acc.z = fmaf(__uint_as_float(f[i][k].z), wt[i][k], acc.z);
                acc.w = fmaf(__uint_as_float(f[i][k].w), wt[i][k], acc.w);
            }
            s1[i].x += acc.x; s1[i].y += acc.y; s1[i].z += acc.z; s1[i].w += acc.w;
            s2[i].x = fmaf(acc.x, acc.x, s2[i].x); s2[i].y = fmaf(acc.y, acc.y, s2[i].y);
            s2[i].z = fmaf(acc.z, acc.z, s2[i].z); s2[i].w = fmaf(acc.w, acc.w, s2[i].w);
        }
    }

    const float inv_nv = 1.0f / (float)(p.V + 1);  // mvsnet.py:135, V there counts the key view
    const size_t plane = (size_t)h * w;
#pragma unroll
    for (int i = 0; i < DPB; ++i) {
        const int d = d0 + i;
        if (d >= D) break;  // block-uniform
        float4 r;
        if constexpr (WARP_ONLY) {
            r = s1[i];
        } else {
            const float mx = s1[i].x * inv_nv, my = s1[i].y * inv_nv, mz = s1[i].z * inv_nv, mw = s1[i].w * inv_nv;
            r = make_float4(fmaf(s2[i].x, inv_nv, -mx * mx), fmaf(s2[i].y, inv_nv, -my * my),
                            fmaf(s2[i].z, inv_nv, -mz * mz), fmaf(s2[i].w, inv_nv, -mw * mw));
        }
        if (p.layout == MVD_LAYOUT_NDHWC) {
#if MVD_K3_EXPERIMENT == 2
            if (active && r.x == 123.456f)
#else
            if (active)
#endif
                *reinterpret_cast<float4*>(p.out + ((((size_t)b * D + d) * h + y) * w + x) * C + q * 4) = r;
            continue;
        }
        // NCDHW: transpose the (pixel, channel) tile through LDS so every channel row is written as
        // PPB consecutive floats.
        __syncthreads();
        stage[(q * 4 + 0) * (PPB + 1) + px] = r.x;
        stage[(q * 4 + 1) * (PPB + 1) + px] = r.y;
        stage[(q * 4 + 2) * (PPB + 1) + px] = r.z;
        stage[(q * 4 + 3) * (PPB + 1) + px] = r.w;
        __syncthreads();
#pragma unroll
        for (int e0 = 0; e0 < C * PPB / 256; ++e0) {
            const int e = tid + e0 * 256;
            const int c = e / PPB, xx = e % PPB;
            if (x0 + xx < w)
                p.out[(((size_t)b * C + c) * D + d) * plane + (size_t)y * w + x0 + xx] = stage[c * (PPB + 1) + xx];
        }
    }
}


// ------------------------------------------------------------------------------------------------
// K3, round-2 form ("located"): C = 32, channel-last output, folded grid arithmetic.
//
// What bounded the kernel above (profiles/r01_k3_uniform_pmc.txt): 273 M vector-ALU wave-instructions per launch
// (the SIMDs 59 % VALU-busy) of which only 85 M are the bilinear/variance FMAs; the rest is the sampling-position
// arithmetic — done by 8 lanes per pixel, i.e. twice per (pixel, plane, view) even with the quad sharing — plus 20
// DPP broadcasts per view, at 148 VGPRs = 3 waves per SIMD, too few to cover the gather latency and the store
// acknowledgements (the store stream by itself runs at 6.6 TB/s: profiles/r02_storebw.txt).
//
// Here a workgroup first LOCATES: thread t computes the position, cell offset and four bilinear weights of exactly
// one (pixel, plane, view) combination per pass (32 pixels x 4 planes x V views = 128 V combinations, no redundancy,
// view wave-uniform so the transform comes through the scalar cache) and parks them in LDS (20 B each).  After one
// barrier the same threads BLEND as before — 8 lanes per pixel, 4 channels each — but read weights and offsets from
// LDS (broadcast reads, no VALU) instead of computing and shuffling them, and keep at most three cells in flight
// (gather_blend_4planes_lds) so that the kernel fits 128 VGPRs = 4 waves per SIMD.  Same arithmetic, operation for
// operation, as warp_variance_kernel: results are bit-identical.

template <int MASK, int I, class TAP>
__device__ __forceinline__ void blend_plane_lds(float4 (&s1)[4], float4 (&s2)[4], const float4 wq, const TAP (&f)[4]) {
    const float w[4] = {wq.x, wq.y, wq.z, wq.w};
    accumulate_cell(s1[I], s2[I], w, f);
}

// taps of one cell: nw at `o`, ne at +128 (folds into the instruction's immediate), sw / se one padded row further
// (the row pitch rides in the scalar offset operand: no per-lane address arithmetic besides `o` itself)
#ifndef MVD_K3_KO
#define MVD_K3_KO 0  // knock-out builds of the marching kernel (tools/ko_k3.sh; wrong results): 2 no gathers, 4 no stores, 1 no locate
#endif
__device__ __forceinline__ void gather_cell_s(u32x4 (&f)[4], __amdgpu_buffer_rsrc_t rsrc, unsigned o, unsigned rowb) {
    if constexpr ((MVD_K3_KO & 2) != 0) {
        f[0] = u32x4{o, o + 1, o + 2, o + 3}; f[1] = u32x4{o + 4, o + 5, o + 6, o + 7};
        f[2] = u32x4{o + rowb, o + 9, o + 10, o + 11}; f[3] = u32x4{o + rowb + 4, o + 13, o + 14, o + 15};
        return;
    }
    f[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);
    f[1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + 128u, 0, 0);
    f[2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, rowb, 0);
    f[3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o + 128u, rowb, 0);
}
// fp16 features: 64 bytes per pixel, 8 per lane
__device__ __forceinline__ void gather_cell_s(u32x2 (&f)[4], __amdgpu_buffer_rsrc_t rsrc, unsigned o, unsigned rowb) {
    f[0] = load_b64(rsrc, o, 0);
    f[1] = load_b64(rsrc, o + 64u, 0);
    f[2] = load_b64(rsrc, o, rowb);
    f[3] = load_b64(rsrc, o + 64u, rowb);
}

template <int MASK, int KO = 0>
__device__ __forceinline__ void gather_blend_4planes_lds(float4 (&s1)[4], float4 (&s2)[4], const float4* __restrict__ wl,
                                                         const unsigned (&off)[4], __amdgpu_buffer_rsrc_t rsrc, unsigned rowb) {
    constexpr bool G1 = MASK & 1, G2 = (MASK >> 1) & 1, G3 = (MASK >> 2) & 1;
    auto gather_cell_s = [](u32x4 (&f)[4], __amdgpu_buffer_rsrc_t r, unsigned o, unsigned rb) {
        if constexpr (KO & 2) {  // knock-out (experiments library only): taps from registers, no memory access
            f[0] = u32x4{o, o + 1, o + 2, o + 3}; f[1] = u32x4{o + 4, o + 5, o + 6, o + 7};
            f[2] = u32x4{o + rb, o + 9, o + 10, o + 11}; f[3] = u32x4{o + rb + 4, o + 13, o + 14, o + 15};
        } else {
            mvd::gather_cell_s(f, r, o, rb);
        }
    };
    u32x4 a[4], b[4], c[4];  // three register sets
    gather_cell_s(a, rsrc, off[0], rowb);
    if constexpr (MASK == 7) {
        gather_cell_s(b, rsrc, off[1], rowb);
        gather_cell_s(c, rsrc, off[2], rowb);
        const float4 w0 = wl[0], w1 = wl[32], w2 = wl[64], w3 = wl[96];  // LDS: lands long before the gathers do
        blend_plane_lds<MASK, 0>(s1, s2, w0, a);
        gather_cell_s(a, rsrc, off[3], rowb);  // into the set plane 0 just released
        blend_plane_lds<MASK, 1>(s1, s2, w1, b);
        blend_plane_lds<MASK, 2>(s1, s2, w2, c);
        blend_plane_lds<MASK, 3>(s1, s2, w3, a);
    } else {
        // at most two of planes 1..3 re-gather: sets b and c take them in order
        if constexpr (G1) gather_cell_s(b, rsrc, off[1], rowb);
        if constexpr (G2) gather_cell_s(G1 ? c : b, rsrc, off[2], rowb);
        if constexpr (G3) gather_cell_s((G1 || G2) ? c : b, rsrc, off[3], rowb);
        const float4 w0 = wl[0], w1 = wl[32], w2 = wl[64], w3 = wl[96];
        blend_plane_lds<MASK, 0>(s1, s2, w0, a);
        const u32x4 (&p1)[4] = G1 ? b : a;
        blend_plane_lds<MASK, 1>(s1, s2, w1, p1);
        const u32x4 (&p2)[4] = G2 ? (G1 ? c : b) : p1;
        blend_plane_lds<MASK, 2>(s1, s2, w2, p2);
        const u32x4 (&p3)[4] = G3 ? ((G1 || G2) ? c : b) : p2;
        blend_plane_lds<MASK, 3>(s1, s2, w3, p3);
    }
}

// KO != 0 only in the experiments library: knock-out builds that time parts of the kernel (they compute wrong results):
// 1 no locate arithmetic, 2 gathers replaced by register values, 4 no stores, 8 one view only
template <int MINW, int KO = 0>
__global__ void __launch_bounds__(256, MINW) warp_variance_located_kernel(WarpParams p) {
    constexpr int DPB = 4, PPB = 32;
    constexpr unsigned PIX = 128;
    extern __shared__ __attribute__((aligned(16))) float4 lds_loc[];  // [V][4][32] float4 weights, then [V][4][32] u32 offsets
    const int V = (KO & 8) ? 1 : p.V;
    unsigned* __restrict__ lds_off = reinterpret_cast<unsigned*>(lds_loc + V * (DPB * PPB));

    const int tid = threadIdx.x;
    const int h = p.h, w = p.w, D = p.D;

    // ---- decode the block index: xcd | (d-chunk fastest, then tile within the XCD's band, then batch) ----
    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    const int dchunks = (D + DPB - 1) / DPB;
    const int dc = j % dchunks; j /= dchunks;
    const int tile_in = j % p.tiles_per_xcd;
    const int b = j / p.tiles_per_xcd;
    const int tile = xcd * p.tiles_per_xcd + tile_in;
    if (tile >= p.tiles_x * h) return;  // block-uniform
    const int y = tile / p.tiles_x;
    const int x0 = (tile - y * p.tiles_x) * PPB;
    const int d0 = dc * DPB;

    const int W2 = w + 3;
    const unsigned rowb = (unsigned)W2 * PIX;             // bytes per padded row
    const unsigned img_bytes = (unsigned)(h + 3) * rowb;  // bytes per padded image

    // ---- locate: one (pixel, plane, view) per thread and pass -------------------------------------------------
    {
        const int lpx = tid & 31, li = (tid >> 5) & 3;
        const int xl = min(x0 + lpx, w - 1);
        const float fx = (float)xl, fy = (float)y;
        const float sx = (float)w / (float)(w - 1), sy = (float)h / (float)(h - 1);
        const float xhi = (float)w, yhi = (float)h;
        const float W2f = (float)W2;
        const float depth = p.depth[(size_t)b * D + min(d0 + li, D - 1)];
        for (int v = __builtin_amdgcn_readfirstlane(tid >> 7); v < ((KO & 1) ? 0 : V); v += 2) {  // wave-uniform view
            const float* __restrict__ M = p.M + ((size_t)v * p.B + b) * 12;       // scalar loads
            const float ax = fmaf(M[0], fx, fmaf(M[1], fy, M[2]));
            const float ay = fmaf(M[4], fx, fmaf(M[5], fy, M[6]));
            const float az = fmaf(M[8], fx, fmaf(M[9], fy, M[10]));
            const float X = fmaf(ax, depth, M[3]), Y = fmaf(ay, depth, M[7]), Z = fmaf(az, depth, M[11]);
            const float rz = __builtin_amdgcn_rcpf(Z);
            float ix = fmaf(X * rz, sx, -0.5f), iy = fmaf(Y * rz, sy, -0.5f);
            ix = __builtin_amdgcn_fmed3f(ix, -1.0f, xhi);
            iy = __builtin_amdgcn_fmed3f(iy, -1.0f, yhi);
            const float xf = floorf(ix), yf = floorf(iy);
            const float wx = ix - xf, wy = iy - yf;
            const float ux = 1.0f - wx, uy = 1.0f - wy;
            const int slot = (v * DPB + li) * PPB + lpx;
            lds_loc[slot] = make_float4(ux * uy, wx * uy, ux * wy, wx * wy);
            lds_off[slot] = (unsigned)(int)fmaf(yf, W2f, xf) * PIX;  // exact in fp32 (checked on the host)
        }
    }

    // ---- blend: 8 lanes per pixel, 4 channels per lane ---------------------------------------------------------
    const int q = tid & 7, px = tid >> 3;
    const int xc = min(x0 + px, w - 1);
    const unsigned org = rowb + PIX + (unsigned)q * 16;  // padded (1,1) + this lane's channel quad
    float4 s1[DPB], s2[DPB];
    {
        const float4 k = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.key) + (size_t)b * img_bytes + org +
                                                          (unsigned)y * rowb + (unsigned)xc * PIX);
        const float4 k2 = make_float4(k.x * k.x, k.y * k.y, k.z * k.z, k.w * k.w);
#pragma unroll
        for (int i = 0; i < DPB; ++i) { s1[i] = k; s2[i] = k2; }
    }
    __syncthreads();

    // the next view's cell offsets and source pointer are fetched (LDS / scalar cache) under the current view's gathers
    unsigned offn[DPB];
    const char* srcn;
    auto fetch_view = [&](int v) {
        const unsigned* __restrict__ ol = lds_off + v * (DPB * PPB) + px;
#pragma unroll
        for (int i = 0; i < DPB; ++i) offn[i] = ol[i * PPB];
        srcn = reinterpret_cast<const char*>(p.src.p[v]);
    };
    fetch_view(0);
    for (int v = 0; v < V; ++v) {
        unsigned off[DPB];
#pragma unroll
        for (int i = 0; i < DPB; ++i) off[i] = offn[i] + org;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(srcn + (size_t)b * img_bytes), 0, (int)img_bytes, 0x00020000);
        const float4* __restrict__ wl = lds_loc + v * (DPB * PPB) + px;
        fetch_view(min(v + 1, V - 1));
        // wave-uniform re-gather pattern: bit i-1 set = some lane's 2x2 cell differs between plane i-1 and plane i
        const unsigned mask = (__builtin_amdgcn_ballot_w64(off[1] != off[0]) != 0 ? 1u : 0u) |
                              (__builtin_amdgcn_ballot_w64(off[2] != off[1]) != 0 ? 2u : 0u) |
                              (__builtin_amdgcn_ballot_w64(off[3] != off[2]) != 0 ? 4u : 0u);
        switch (mask) {
            case 0: gather_blend_4planes_lds<0, KO>(s1, s2, wl, off, rsrc, rowb); break;
            case 1: gather_blend_4planes_lds<1, KO>(s1, s2, wl, off, rsrc, rowb); break;
            case 2: gather_blend_4planes_lds<2, KO>(s1, s2, wl, off, rsrc, rowb); break;
            case 3: gather_blend_4planes_lds<3, KO>(s1, s2, wl, off, rsrc, rowb); break;
            case 4: gather_blend_4planes_lds<4, KO>(s1, s2, wl, off, rsrc, rowb); break;
            case 5: gather_blend_4planes_lds<5, KO>(s1, s2, wl, off, rsrc, rowb); break;
            case 6: gather_blend_4planes_lds<6, KO>(s1, s2, wl, off, rsrc, rowb); break;
            default: gather_blend_4planes_lds<7, KO>(s1, s2, wl, off, rsrc, rowb); break;
        }
    }

    const float inv_nv = 1.0f / (float)(p.V + 1);  // mvsnet.py:135, V there counts the key view
    if (x0 + px < w) {
#pragma unroll
        for (int i = 0; i < DPB; ++i) {
            const int d = d0 + i;
            if (d >= D) break;  // block-uniform
            const float mx = s1[i].x * inv_nv, my = s1[i].y * inv_nv, mz = s1[i].z * inv_nv, mw = s1[i].w * inv_nv;
            const float4 r = make_float4(fmaf(s2[i].x, inv_nv, -mx * mx), fmaf(s2[i].y, inv_nv, -my * my),
                                         fmaf(s2[i].z, inv_nv, -mz * mz), fmaf(s2[i].w, inv_nv, -mw * mw));
            if ((KO & 4) && r.x != 123.456f) continue;
            *reinterpret_cast<float4*>(p.out + ((((size_t)b * D + d) * h + y) * w + (x0 + px)) * 32 + q * 4) = r;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Marching form of the located kernel.  Knock-out timings of warp_variance_located_kernel (profiles/r02_k3_located_ko.txt)
// show that nothing in it overlaps: 0.16 ms of per-workgroup prologue latency (kernel arguments -> depth -> locate ->
// barrier), 0.10 ms of blend arithmetic per view, +0.15 ms of exposed gather latency, +0.17 ms of exposed store
// acknowledgements add up linearly to the 0.82 ms.  Here a workgroup keeps its 32-pixel row segment and MARCHES through
// `nch` consecutive 4-plane chunks: index decode and key fetch once, chunk c+1 is located (into the other half of a
// double-buffered LDS table, depths through the scalar cache so that nothing queues behind the stores) before chunk c is
// blended, one barrier per chunk, and the stores of chunk c drain while chunk c+1 is located and its first gathers fly.
// NSETS = 2 keeps two cells in flight instead of three (96 VGPRs = 5 waves per SIMD).
constexpr int cell_of(int mask, int i) { return i == 0 ? 0 : cell_of(mask, i - 1) + ((mask >> (i - 1)) & 1); }
constexpr int ncells_of(int mask) { return cell_of(mask, 3) + 1; }
constexpr int first_plane_of_cell(int mask, int k) {
    for (int i = 0; i < 4; ++i)
        if (cell_of(mask, i) == k) return i;
    return 3;
}

template <int MASK, int K, int I, class TAP>
__device__ __forceinline__ void blend_if_cell(float4 (&s1)[4], float4 (&s2)[4], const float4 (&w)[4], const TAP (&X)[4],
                                              const TAP (&Y)[4]) {
    if constexpr (cell_of(MASK, I) == K) blend_plane_lds<MASK, I>(s1, s2, w[I], (K & 1) ? Y : X);
}

struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};

// `after_last_gather` runs right behind the LAST gather request of the view (the parked stores of the previous chunk go there:
// every later wait of this view is for loads that are OLDER than those stores)
template <int MASK, int K, class TAP, class Hook = NoHook>
__device__ __forceinline__ void cell_step(float4 (&s1)[4], float4 (&s2)[4], const float4 (&w)[4], TAP (&X)[4], TAP (&Y)[4],
                                          const unsigned (&off)[4], __amdgpu_buffer_rsrc_t rsrc, unsigned rowb,
                                          const Hook& after_last_gather = Hook()) {
    if constexpr (K < ncells_of(MASK)) {
        blend_if_cell<MASK, K, 0>(s1, s2, w, X, Y);
        blend_if_cell<MASK, K, 1>(s1, s2, w, X, Y);
        blend_if_cell<MASK, K, 2>(s1, s2, w, X, Y);
        blend_if_cell<MASK, K, 3>(s1, s2, w, X, Y);
        if constexpr (K + 2 < ncells_of(MASK)) {  // the set this cell just released takes the cell after next
            __builtin_amdgcn_sched_barrier(0);    // (left alone, the scheduler hoists these loads above the blends and spills)
            gather_cell_s((K & 1) ? Y : X, rsrc, off[first_plane_of_cell(MASK, K + 2)], rowb);
            if constexpr (K + 2 == ncells_of(MASK) - 1) {
                after_last_gather();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

template <int MASK, class TAP = u32x4, class Hook = NoHook>
__device__ __forceinline__ void gather_blend_4planes_2sets(float4 (&s1)[4], float4 (&s2)[4], const float4* __restrict__ wl,
                                                           const unsigned (&off)[4], __amdgpu_buffer_rsrc_t rsrc, unsigned rowb,
                                                           const Hook& after_last_gather = Hook()) {
    TAP X[4], Y[4];
    gather_cell_s(X, rsrc, off[0], rowb);
    if constexpr (ncells_of(MASK) > 1) gather_cell_s(Y, rsrc, off[first_plane_of_cell(MASK, 1)], rowb);
    if constexpr (ncells_of(MASK) <= 2) {
        after_last_gather();
        __builtin_amdgcn_sched_barrier(0);
    }
    const float4 w[4] = {wl[0], wl[32], wl[64], wl[96]};
    cell_step<MASK, 0>(s1, s2, w, X, Y, off, rsrc, rowb, after_last_gather);
    cell_step<MASK, 1>(s1, s2, w, X, Y, off, rsrc, rowb, after_last_gather);
    cell_step<MASK, 2>(s1, s2, w, X, Y, off, rsrc, rowb, after_last_gather);
    cell_step<MASK, 3>(s1, s2, w, X, Y, off, rsrc, rowb, after_last_gather);
}

// Pipelined form: the first cell of this view (set X) was gathered while the PREVIOUS view was blended; the second cell (if
// the chunk has one) is requested first thing, then the chain of cell_steps runs as above.
template <int MASK, class TAP = u32x4>
__device__ __forceinline__ void blend_view_prefetched(float4 (&s1)[4], float4 (&s2)[4], const float4* __restrict__ wl, TAP (&X)[4],
                                                      TAP (&Y)[4], const unsigned (&off)[4], __amdgpu_buffer_rsrc_t rsrc,
                                                      unsigned rowb) {
    if constexpr (ncells_of(MASK) > 1) gather_cell_s(Y, rsrc, off[first_plane_of_cell(MASK, 1)], rowb);
    const float4 w[4] = {wl[0], wl[32], wl[64], wl[96]};
    cell_step<MASK, 0>(s1, s2, w, X, Y, off, rsrc, rowb);
    cell_step<MASK, 1>(s1, s2, w, X, Y, off, rsrc, rowb);
    cell_step<MASK, 2>(s1, s2, w, X, Y, off, rsrc, rowb);
    cell_step<MASK, 3>(s1, s2, w, X, Y, off, rsrc, rowb);
}

// F16: features are fp16 zero-bordered channel-last maps (64 B per pixel), the volume is written as fp16 (B,D,h,w,32);
// positions, weights, blend and variance stay fp32 (mvd_warp_variance_f16, BASELINE configs[3]).
// WP (wave-private locate): every wave locates the 128 (pixel, plane, view) combinations of ITS OWN 8 pixels (2 per lane;
// the two half-waves take even / odd views, so the transforms come from a small LDS table instead of the scalar cache) and
// is the only reader of those table entries: no workgroup barrier inside the march, the four waves drift apart freely.
// PIPE: views are software-pipelined.  A wave spends most of a (chunk, view) waiting for the view's first gather (vector ALU 46 %
// busy, texture addresser 74 %, four waves per SIMD: profiles/r02_k3_march_pmc.txt); here the first cell of view v+1 is
// requested before view v is blended, into a second pair of tap sets (64 tap VGPRs, three waves per SIMD).
// PARK (needs WP): a chunk's results are not stored at its end but parked in LDS (16 KB, wave-private) and stored from the
// middle of the NEXT chunk's first view, right behind that view's last gather request.  vmcnt retires loads and stores in one
// order: stores issued at the end of a chunk sit in front of the next chunk's first gathers, and the first blend then waits for
// their write acknowledgements (knock-out timings, profiles/r02_k3_march_ko.txt: no stores -0.10 ms, no gathers -0.12 ms,
// neither -0.25 ms of 0.72).
template <int MINW, int NSETS, bool F16 = false, bool WP = false, bool PIPE = false, bool PARK = false>
__global__ void __launch_bounds__(256, MINW) warp_variance_march_kernel(WarpParams p, int nch) {
    constexpr int DPB = 4, PPB = 32;
    constexpr unsigned PIX = F16 ? 64 : 128;  // bytes per pixel
    constexpr unsigned QB = F16 ? 8 : 16;     // bytes per lane (4 channels)
    extern __shared__ __attribute__((aligned(16))) float4 lds_raw[];  // 2 x ([V][4][32] float4 weights + [V][4][32] u32 offsets)
    const int V = p.V;
    const int half_q = V * (DPB * PPB) * 5 / 4;  // float4 slots per table half (weights + offsets)
    constexpr int NH = (WP && PARK) ? 1 : 2;     // table halves in the allocation (the wave-private locate uses one)

    const int tid = threadIdx.x;
    const int h = p.h, w = p.w, D = p.D;

    // ---- decode the block index: xcd | (chunk group fastest, then tile within the XCD's band, then batch) ----
    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    const int dchunks = (D + DPB - 1) / DPB;
    const int dgroups = (dchunks + nch - 1) / nch;
    const int dg = j % dgroups; j /= dgroups;
    const int tile_in = j % p.tiles_per_xcd;
    const int b = j / p.tiles_per_xcd;
    const int tile = xcd * p.tiles_per_xcd + tile_in;
    if (tile >= p.tiles_x * h) return;  // block-uniform
    const int y = tile / p.tiles_x;
    const int x0 = (tile - y * p.tiles_x) * PPB;
    const int c_begin = dg * nch, c_end = min(c_begin + nch, dchunks);

    const int W2 = w + 3;
    const unsigned rowb = (unsigned)W2 * PIX;             // bytes per padded row
    const unsigned img_bytes = (unsigned)(h + 3) * rowb;  // bytes per padded image
    // constant address space: uniform reads of the depth samples and transforms become scalar-cache loads (lgkmcnt);
    // as ordinary global loads they would be vector-memory operations that retire in order BEHIND the previous chunk's
    // stores (profiles/r02_k3_march_pmc.txt: 17 vector loads per wave too many)
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat* dvals = (cfloat*)(p.depth + (size_t)b * D);

    // locate-phase constants: thread = (pixel lpx, plane li), views two at a time (wave-uniform)
    const int lpx = tid & 31, li = (tid >> 5) & 3;
    const float lfx = (float)min(x0 + lpx, w - 1), lfy = (float)y;
    const float sx = (float)w / (float)(w - 1), sy = (float)h / (float)(h - 1);
    const float xhi = (float)w, yhi = (float)h;
    const float W2f = (float)W2;
    const int v_first = __builtin_amdgcn_readfirstlane(tid >> 7);
    // wave-private form: lane = (pixel of this wave l&7, plane (l>>3)&3, view parity l>>5)
    float4* __restrict__ mtab = lds_raw + NH * half_q + 256;  // [V][3] float4: the composed transforms (WP only)
    if constexpr (WP) {
        if (tid < V * 3) {
            cfloat* Mg = (cfloat*)(p.M + ((size_t)(tid / 3) * p.B + b) * 12 + (tid % 3) * 4);
            mtab[tid] = make_float4(Mg[0], Mg[1], Mg[2], Mg[3]);
        }
    }
    auto locate_wp = [&](int c) {
        float4* __restrict__ loc = lds_raw;
        unsigned* __restrict__ offs = reinterpret_cast<unsigned*>(loc + V * (DPB * PPB));
        // lane-derived values are recomputed from an opaque copy of the thread index (a few integer ops per chunk) instead of
        // living in registers across the whole march: the kernel sits exactly at the 128-VGPR boundary of 4 waves per SIMD
        int t_ = tid;
        asm volatile("" : "+v"(t_));
        const int wpx = (t_ >> 6) * 8 + (t_ & 7), wpi = (t_ >> 3) & 3, wvp = (t_ >> 5) & 1;
        const float wfx = (float)min(x0 + wpx, w - 1);
        const int d0 = c * DPB;
        const float e0 = dvals[min(d0, D - 1)], e1 = dvals[min(d0 + 1, D - 1)], e2 = dvals[min(d0 + 2, D - 1)],
                    e3 = dvals[min(d0 + 3, D - 1)];
        const float depth = wpi == 0 ? e0 : wpi == 1 ? e1 : wpi == 2 ? e2 : e3;
        for (int v = wvp; v < V; v += 2) {
            const float4 m0 = mtab[v * 3], m1 = mtab[v * 3 + 1], m2 = mtab[v * 3 + 2];
            const float ax = fmaf(m0.x, wfx, fmaf(m0.y, lfy, m0.z));
            const float ay = fmaf(m1.x, wfx, fmaf(m1.y, lfy, m1.z));
            const float az = fmaf(m2.x, wfx, fmaf(m2.y, lfy, m2.z));
            const float X = fmaf(ax, depth, m0.w), Y = fmaf(ay, depth, m1.w), Z = fmaf(az, depth, m2.w);
            const float rz = __builtin_amdgcn_rcpf(Z);
            float ix = fmaf(X * rz, sx, -0.5f), iy = fmaf(Y * rz, sy, -0.5f);
            ix = __builtin_amdgcn_fmed3f(ix, -1.0f, xhi);
            iy = __builtin_amdgcn_fmed3f(iy, -1.0f, yhi);
            const float xf = floorf(ix), yf = floorf(iy);
            const float wx = ix - xf, wy = iy - yf;
            const float ux = 1.0f - wx, uy = 1.0f - wy;
            const int slot = (v * DPB + wpi) * PPB + wpx;
            loc[slot] = make_float4(ux * uy, wx * uy, ux * wy, wx * wy);
            offs[slot] = (unsigned)(int)fmaf(yf, W2f, xf) * PIX;
        }
    };
    auto locate = [&](int c, int buf) {
        float4* __restrict__ loc = lds_raw + buf * half_q;
        unsigned* __restrict__ offs = reinterpret_cast<unsigned*>(loc + V * (DPB * PPB));
        // the chunk's four depths are wave-uniform: scalar loads (lgkmcnt), so nothing here queues behind the
        // vector-memory stores of the previous chunk
        const int d0 = c * DPB;
        const float e0 = dvals[min(d0, D - 1)], e1 = dvals[min(d0 + 1, D - 1)], e2 = dvals[min(d0 + 2, D - 1)],
                    e3 = dvals[min(d0 + 3, D - 1)];
        const float depth = li == 0 ? e0 : li == 1 ? e1 : li == 2 ? e2 : e3;
        for (int v = v_first; v < ((MVD_K3_KO & 1) ? 0 : V); v += 2) {
            cfloat* M = (cfloat*)(p.M + ((size_t)v * p.B + b) * 12);  // scalar loads
            const float ax = fmaf(M[0], lfx, fmaf(M[1], lfy, M[2]));
            const float ay = fmaf(M[4], lfx, fmaf(M[5], lfy, M[6]));
            const float az = fmaf(M[8], lfx, fmaf(M[9], lfy, M[10]));
            const float X = fmaf(ax, depth, M[3]), Y = fmaf(ay, depth, M[7]), Z = fmaf(az, depth, M[11]);
            const float rz = __builtin_amdgcn_rcpf(Z);
            float ix = fmaf(X * rz, sx, -0.5f), iy = fmaf(Y * rz, sy, -0.5f);
            ix = __builtin_amdgcn_fmed3f(ix, -1.0f, xhi);
            iy = __builtin_amdgcn_fmed3f(iy, -1.0f, yhi);
            const float xf = floorf(ix), yf = floorf(iy);
            const float wx = ix - xf, wy = iy - yf;
            const float ux = 1.0f - wx, uy = 1.0f - wy;
            const int slot = (v * DPB + li) * PPB + lpx;
            loc[slot] = make_float4(ux * uy, wx * uy, ux * wy, wx * wy);
            offs[slot] = (unsigned)(int)fmaf(yf, W2f, xf) * PIX;  // exact in fp32 (checked on the host)
        }
    };

    // blend-phase constants: 8 lanes per pixel, 4 channels per lane
    const int q = tid & 7, px = tid >> 3;
    const int xc = min(x0 + px, w - 1);
    const unsigned org = rowb + PIX + (unsigned)q * QB;  // padded (1,1) + this lane's channel quad
    // the key features of this thread's (pixel, channel quad) stay in LDS between chunks (4 fewer long-lived VGPRs)
    float4* __restrict__ key_slot = lds_raw + NH * half_q + tid;
    {
        const char* kp = reinterpret_cast<const char*>(p.key) + (size_t)b * img_bytes + org + (unsigned)y * rowb + (unsigned)xc * PIX;
        if constexpr (F16) {
            const u32x2 kh = *reinterpret_cast<const u32x2*>(kp);
            const f16x2 lo = as_h2(kh.x), hi = as_h2(kh.y);
            *key_slot = make_float4((float)lo.x, (float)lo.y, (float)hi.x, (float)hi.y);
        } else {
            *key_slot = *reinterpret_cast<const float4*>(kp);
        }
    }
    const float inv_nv = 1.0f / (float)(V + 1);  // mvsnet.py:135, V there counts the key view
    // stores: one descriptor per output plane (scalar arithmetic), one 32-bit offset per lane.  Inactive lanes (ragged
    // right edge) carry pixel w-1 like the last active lane and store the same values to the same address: no divergent
    // branch around the stores
    const unsigned out_off = ((unsigned)y * (unsigned)w + (unsigned)xc) * PIX + (unsigned)q * QB;
    const size_t plane_bytes = (size_t)h * w * PIX;
    // PARK: [4 planes][256 lanes] float4 behind the transforms; each lane reads back what it wrote
    float4* __restrict__ park_base = lds_raw + NH * half_q + 256 + ((V * 3 + 3) & ~3);
    int parked_d0 = -1;  // first plane of the chunk whose results are parked (wave-uniform)
    auto drain = [&]() {
        if (parked_d0 < 0) return;
        // the kernel sits exactly at 128 VGPRs: lane-derived addresses are recomputed from an opaque copy of the thread index
        // and the planes leave one at a time (4 VGPRs in flight)
        int t_ = tid;
        asm volatile("" : "+v"(t_));
        const unsigned oo = ((unsigned)y * (unsigned)w + (unsigned)min(x0 + (t_ >> 3), w - 1)) * PIX + (unsigned)(t_ & 7) * QB;
        const float4* __restrict__ pk = park_base + t_;
#pragma unroll
        for (int i = 0; i < DPB; ++i) {
            const float4 r = pk[i * 256];
            const int d = min(parked_d0 + i, D - 1);
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<char*>(p.out) + ((size_t)b * D + d) * plane_bytes, 0, parked_d0 + i < D ? (int)plane_bytes : 0, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(r.x), __float_as_uint(r.y), __float_as_uint(r.z),
                                                         __float_as_uint(r.w)}, orsrc, oo, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if constexpr (WP) __syncthreads();  // transforms and key slots are staged; no further workgroup barrier
    else locate(c_begin, 0);
    int buf = 0;
    for (int c = c_begin; c < c_end; ++c, buf ^= (WP ? 0 : 1)) {
        if constexpr (WP) {
            locate_wp(c);  // this wave's own table entries (same-wave LDS accesses execute in order)
        } else {
            __syncthreads();  // table `buf` is complete; every wave is done reading table `buf ^ 1`
            if (c + 1 < c_end) locate(c + 1, buf ^ 1);
        }
        const float4* __restrict__ loc = lds_raw + buf * half_q;
        const unsigned* __restrict__ offs = reinterpret_cast<const unsigned*>(loc + V * (DPB * PPB));
        float4 s1[DPB], s2[DPB];
        {
            const float4 k = *key_slot;
            const float4 k2 = make_float4(k.x * k.x, k.y * k.y, k.z * k.z, k.w * k.w);
#pragma unroll
            for (int i = 0; i < DPB; ++i) { s1[i] = k; s2[i] = k2; }
        }
        if constexpr (PIPE) {
            using TAP = typename std::conditional<F16, u32x2, u32x4>::type;
            TAP X0[4], Y0[4], X1[4], Y1[4];
            unsigned offa[DPB], offb[DPB], ma = 0, mb = 0;
            __amdgpu_buffer_rsrc_t ra, rb;
            // offsets + re-gather pattern of view v (wave-uniform mask), its descriptor, and the request for its first cell
            auto open_view = [&](int v, unsigned (&off)[DPB], unsigned& mask, TAP (&X)[4]) {
                const unsigned* __restrict__ ol = offs + v * (DPB * PPB) + px;
#pragma unroll
                for (int i = 0; i < DPB; ++i) off[i] = ol[i * PPB] + org;
                mask = (__builtin_amdgcn_ballot_w64(off[1] != off[0]) != 0 ? 1u : 0u) |
                       (__builtin_amdgcn_ballot_w64(off[2] != off[1]) != 0 ? 2u : 0u) |
                       (__builtin_amdgcn_ballot_w64(off[3] != off[2]) != 0 ? 4u : 0u);
                const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<char*>(reinterpret_cast<const char*>(p.src.p[v]) + (size_t)b * img_bytes), 0, (int)img_bytes, 0x00020000);
                gather_cell_s(X, r, off[0], rowb);
                return r;
            };
#define MVD_BLEND(Mk, Xs, Ys, offs_, rs_, v_) \
    case Mk: blend_view_prefetched<Mk, TAP>(s1, s2, loc + (v_) * (DPB * PPB) + px, Xs, Ys, offs_, rs_, rowb); break;
#define MVD_BLEND_ALL(mask_, Xs, Ys, offs_, rs_, v_)                                                                      \
    switch (mask_) {                                                                                                     \
        MVD_BLEND(0, Xs, Ys, offs_, rs_, v_) MVD_BLEND(1, Xs, Ys, offs_, rs_, v_) MVD_BLEND(2, Xs, Ys, offs_, rs_, v_)  \
        MVD_BLEND(3, Xs, Ys, offs_, rs_, v_) MVD_BLEND(4, Xs, Ys, offs_, rs_, v_) MVD_BLEND(5, Xs, Ys, offs_, rs_, v_)  \
        MVD_BLEND(6, Xs, Ys, offs_, rs_, v_)                                                                             \
        default: blend_view_prefetched<7, TAP>(s1, s2, loc + (v_) * (DPB * PPB) + px, Xs, Ys, offs_, rs_, rowb); break;   \
    }
            // Every path between a request and the first use of its taps is straight-line code: behind a conditional request
            // the compiler must assume the smaller number of outstanding loads and would wait for the prefetch it just issued.
            ra = open_view(0, offa, ma, X0);
            for (int v = 0; v + 1 < V; v += 2) {
                rb = open_view(v + 1, offb, mb, X1);
                MVD_BLEND_ALL(ma, X0, Y0, offa, ra, v)
                if (v + 2 < V) {
                    ra = open_view(v + 2, offa, ma, X0);
                    MVD_BLEND_ALL(mb, X1, Y1, offb, rb, v + 1)
                } else {
                    MVD_BLEND_ALL(mb, X1, Y1, offb, rb, v + 1)
                }
            }
            if (V & 1) {  // the last view of an odd count was opened by the iteration before it (or above, V = 1)
                MVD_BLEND_ALL(ma, X0, Y0, offa, ra, V - 1)
            }
#undef MVD_BLEND_ALL
#undef MVD_BLEND
        } else {
            unsigned offn[DPB];
            const char* srcn;
            auto fetch_view = [&](int v) {
                const unsigned* __restrict__ ol = offs + v * (DPB * PPB) + px;
    #pragma unroll
                for (int i = 0; i < DPB; ++i) offn[i] = ol[i * PPB];
                srcn = reinterpret_cast<const char*>(p.src.p[v]);
            };
            fetch_view(0);
            for (int v = 0; v < V; ++v) {
                unsigned off[DPB];
    #pragma unroll
                for (int i = 0; i < DPB; ++i) off[i] = offn[i] + org;
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<char*>(srcn + (size_t)b * img_bytes), 0, (int)img_bytes, 0x00020000);
                const float4* __restrict__ wl = loc + v * (DPB * PPB) + px;
                fetch_view(min(v + 1, V - 1));
                const unsigned mask = (__builtin_amdgcn_ballot_w64(off[1] != off[0]) != 0 ? 1u : 0u) |
                                      (__builtin_amdgcn_ballot_w64(off[2] != off[1]) != 0 ? 2u : 0u) |
                                      (__builtin_amdgcn_ballot_w64(off[3] != off[2]) != 0 ? 4u : 0u);
    #define MVD_CASE(Mk)                                                                                   \
        if constexpr (F16) gather_blend_4planes_2sets<Mk, u32x2>(s1, s2, wl, off, rsrc, rowb);             \
        else if constexpr (NSETS == 2) gather_blend_4planes_2sets<Mk, u32x4>(s1, s2, wl, off, rsrc, rowb); \
        else gather_blend_4planes_lds<Mk>(s1, s2, wl, off, rsrc, rowb);                                   \
        break;
    #define MVD_CASE_HOOK(Mk) gather_blend_4planes_2sets<Mk, u32x4>(s1, s2, wl, off, rsrc, rowb, drain_v0); break;
                if constexpr (PARK) {  // the previous chunk's parked results leave behind the first view's last gather request
                    const bool first = v == 0;  // wave-uniform
                    auto drain_v0 = [&]() { if (first) drain(); };
                    switch (mask) {
                        case 0: MVD_CASE_HOOK(0)
                        case 1: MVD_CASE_HOOK(1)
                        case 2: MVD_CASE_HOOK(2)
                        case 3: MVD_CASE_HOOK(3)
                        case 4: MVD_CASE_HOOK(4)
                        case 5: MVD_CASE_HOOK(5)
                        case 6: MVD_CASE_HOOK(6)
                        default: MVD_CASE_HOOK(7)
                    }
                } else {
                    switch (mask) {
                        case 0: MVD_CASE(0)
                        case 1: MVD_CASE(1)
                        case 2: MVD_CASE(2)
                        case 3: MVD_CASE(3)
                        case 4: MVD_CASE(4)
                        case 5: MVD_CASE(5)
                        case 6: MVD_CASE(6)
                        default: MVD_CASE(7)
                    }
                }
    #undef MVD_CASE_HOOK
    #undef MVD_CASE
            }
        }
        const int d0 = c * DPB;
#pragma unroll
        for (int i = 0; i < DPB; ++i) {
            if (d0 + i >= D) break;  // block-uniform (only in the last chunk of a D that is not a multiple of 4)
            const float mx = s1[i].x * inv_nv, my = s1[i].y * inv_nv, mz = s1[i].z * inv_nv, mw = s1[i].w * inv_nv;
            const float4 r = make_float4(fmaf(s2[i].x, inv_nv, -mx * mx), fmaf(s2[i].y, inv_nv, -my * my),
                                         fmaf(s2[i].z, inv_nv, -mz * mz), fmaf(s2[i].w, inv_nv, -mw * mw));
            if constexpr (PARK) {
                park_base[i * 256 + tid] = r;
                continue;
            }
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<char*>(p.out) + ((size_t)b * D + d0 + i) * plane_bytes, 0, (int)plane_bytes, 0x00020000);
            if constexpr (F16) {  // round to nearest even, one rounding
                const f16x2 lo = {(_Float16)r.x, (_Float16)r.y}, hi = {(_Float16)r.z, (_Float16)r.w};
                store_b64(u32x2{as_u32(lo), as_u32(hi)}, orsrc, out_off);
            } else {
                if ((MVD_K3_KO & 4) == 0 || r.x == 123.456f)
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(r.x), __float_as_uint(r.y), __float_as_uint(r.z),
                                                                 __float_as_uint(r.w)}, orsrc, out_off, 0, 0);
            }
        }
        if constexpr (PARK) parked_d0 = d0;
    }
    if constexpr (PARK) drain();
}

// LDS of the marching kernel grows with V (the tile kernel's does not): up to 64 KiB it launches without an attribute and
// keeps at least two workgroups per CU; beyond that (V >= 13) callers take a kernel whose LDS does not depend on V
static size_t march_lds_bytes(int V) {
    return 2 * (size_t)V * 4 * 32 * (sizeof(float4) + sizeof(unsigned)) + 256 * sizeof(float4) + (size_t)V * 3 * sizeof(float4);
}
static bool march_fits(int V) { return march_lds_bytes(V) <= 64 * 1024; }

static int launch_warp_march(const WarpParams& p0, hipStream_t st, int minw, int nsets, int nch, bool f16 = false) {
    WarpParams p = p0;
    p.tiles_x = (p.w + 31) / 32;
    const long long tiles = (long long)p.tiles_x * p.h;
    p.tiles_per_xcd = (int)((tiles + 7) / 8);
    const int dchunks = (p.D + 3) / 4;
    if (nch < 1) nch = 1;
    if ((long long)p.h * p.w * (f16 ? 64 : 128) >= 0x7fffffffLL) {  // one output plane is addressed through a 32-bit buffer offset
        set_error("warp_variance: an output plane of %dx%dx32 floats exceeds the 2 GiB buffer-offset range", p.h, p.w);
        return MVD_ERR_INVALID_ARG;
    }
    const int dgroups = (dchunks + nch - 1) / nch;
    const long long nblk = 8LL * p.tiles_per_xcd * dgroups * p.B;
    if (nblk > 0x7fffffffLL) {
        set_error("warp_variance: %lld workgroups exceed the grid limit", nblk);
        return MVD_ERR_INVALID_ARG;
    }
    const size_t lds = march_lds_bytes(p.V);  // 5 KiB per view + key slots + transforms (callers check march_fits)
    const dim3 grid((unsigned)nblk);
    timing_begin(st);
#define MVD_M(MW, NS) hipLaunchKernelGGL((warp_variance_march_kernel<MW, NS>), grid, dim3(256), lds, st, p, nch)
    if (f16) {
        hipLaunchKernelGGL((warp_variance_march_kernel<4, 2, true>), grid, dim3(256), lds, st, p, nch);
        timing_end(st);
        return launch_status("warp_variance_march_f16");
    }
    switch (minw * 10 + nsets) {
#ifdef MVD_EXPERIMENTS
        case 43: MVD_M(4, 3); break;
        case 52: MVD_M(5, 2); break;
        case 62: MVD_M(6, 2); break;
        case 72: hipLaunchKernelGGL((warp_variance_march_kernel<4, 2, false, true>), grid, dim3(256), lds, st, p, nch); break;  // "M7,2,n": wave-private locate
        case 82: hipLaunchKernelGGL((warp_variance_march_kernel<3, 2, false, false, true>), grid, dim3(256), lds, st, p, nch); break;  // "M8,2,n": views pipelined
        case 92: hipLaunchKernelGGL((warp_variance_march_kernel<4, 2, false, false, true>), grid, dim3(256), lds, st, p, nch); break;  // "M9,2,n": views pipelined, 128 VGPRs
        case 102: hipLaunchKernelGGL((warp_variance_march_kernel<4, 2, false, true, false, true>), grid, dim3(256), lds + 4 * 256 * sizeof(float4) + 64, st, p, nch); break;  // "M10,2,n": wave-private locate + parked stores
        case 112: hipLaunchKernelGGL((warp_variance_march_kernel<3, 2, false, true, false, true>), grid, dim3(256), lds + 4 * 256 * sizeof(float4) + 64, st, p, nch); break;  // "M11,2,n": the same at three waves per SIMD
#endif
        default: MVD_M(4, 2); break;
    }
#undef MVD_M
    timing_end(st);
    return launch_status("warp_variance_march");
}

static int launch_warp_located(const WarpParams& p0, hipStream_t st, int minw, int ko = 0) {
    WarpParams p = p0;
    p.tiles_x = (p.w + 31) / 32;
    const long long tiles = (long long)p.tiles_x * p.h;
    p.tiles_per_xcd = (int)((tiles + 7) / 8);
    const long long nblk = 8LL * p.tiles_per_xcd * ((p.D + 3) / 4) * p.B;
    if (nblk > 0x7fffffffLL) {
        set_error("warp_variance: %lld workgroups exceed the grid limit", nblk);
        return MVD_ERR_INVALID_ARG;
    }
    const size_t lds = (size_t)p.V * 4 * 32 * (sizeof(float4) + sizeof(unsigned));  // 2.5 KiB per view
    timing_begin(st);
#ifdef MVD_EXPERIMENTS
    switch (ko) {
#define MVD_KO(K) case K: hipLaunchKernelGGL((warp_variance_located_kernel<4, K>), dim3((unsigned)nblk), dim3(256), lds, st, p); break;
        MVD_KO(1) MVD_KO(2) MVD_KO(4) MVD_KO(6) MVD_KO(7) MVD_KO(8) MVD_KO(14) MVD_KO(15)
#undef MVD_KO
        default: break;
    }
    if (ko == 0 && minw == 3) hipLaunchKernelGGL(warp_variance_located_kernel<3>, dim3((unsigned)nblk), dim3(256), lds, st, p);
    if (ko == 0 && minw != 3)
#endif
    hipLaunchKernelGGL(warp_variance_located_kernel<4>, dim3((unsigned)nblk), dim3(256), lds, st, p);
    timing_end(st);
    return launch_status("warp_variance_located");
}

// (planes per workgroup, min waves per SIMD) — tuned on MI355X; MVD_K3_CFG="dpb,minw"
// selects another compiled variant for experiments (C = 32 only).
static void warp_cfg(int& dpb, int& minw, int& reuse) {
    dpb = 4; minw = 3; reuse = 2;
    if (const char* e = exp_env("MVD_K3_CFG")) {
        if (e[0] >= '0' && e[0] <= '9') { sscanf(e, "%d,%d", &dpb, &minw); reuse = 0; }
        if (e[0] == 'r') { sscanf(e, "r%d,%d", &dpb, &minw); reuse = 1; }
        if (e[0] == 'u') { sscanf(e, "u%d,%d", &dpb, &minw); reuse = 2; }
        if (e[0] == 'v') { sscanf(e, "v%d,%d", &dpb, &minw); reuse = 4; }
    }
}

static size_t padded_image_floats(int C, int h, int w) { return (size_t)(h + 3) * (w + 3) * C; }
static size_t padded_slot_bytes(int B, int C, int h, int w) {
    return align_up((size_t)B * padded_image_floats(C, h, w) * sizeof(float), 256);
}

// clears the slot and writes the zero-bordered channel-last copy of one (B,C,h,w) feature map into it
static int repack_padded(const float* src, float* dst, int B, int C, int h, int w, hipStream_t st) {
    if (hipMemsetAsync(dst, 0, padded_slot_bytes(B, C, h, w), st) != hipSuccess) return launch_status("repack: memset");
    const long long hw = (long long)h * w;
    dim3 grid((unsigned)((hw + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B);
    hipLaunchKernelGGL(repack_padded_kernel, grid, dim3(256), 0, st, src, dst, C, h, w);
    return launch_status("repack_padded");
}

int repack_padded_launch(const float* src, float* dst, int B, int C, int h, int w, hipStream_t st) {
    return repack_padded(src, dst, B, C, h, w, st);
}
size_t padded_slot_bytes_public(int B, int C, int h, int w) { return padded_slot_bytes(B, C, h, w); }

template <bool WARP_ONLY>
static int launch_warp(const WarpParams& p0, int C, hipStream_t st) {
    WarpParams p = p0;
    const int lpp = C / 4;
    int dpb, minw, reuse;
    warp_cfg(dpb, minw, reuse);
    if (lpp != 8 || p.exact_grid) { dpb = 4; minw = 3; reuse = 0; }  // other channel counts, exact grids: the plain variant
    const int ppb = 256 / lpp;
    p.tiles_x = (p.w + ppb - 1) / ppb;
    const long long tiles = (long long)p.tiles_x * p.h;
    p.tiles_per_xcd = (int)((tiles + 7) / 8);
    const long long nblk = 8LL * p.tiles_per_xcd * ((p.D + dpb - 1) / dpb) * p.B;
    if (nblk > 0x7fffffffLL) {
        set_error("warp_variance: %lld workgroups exceed the grid limit", nblk);
        return MVD_ERR_INVALID_ARG;
    }
    const dim3 grid((unsigned)nblk);
    timing_begin(st);
#define MVD_LAUNCH(L, DPB, MW) hipLaunchKernelGGL((warp_variance_kernel<L, WARP_ONLY, DPB, MW>), grid, dim3(256), 0, st, p)
#define MVD_LAUNCH_R(L, DPB, MW) hipLaunchKernelGGL((warp_variance_kernel<L, WARP_ONLY, DPB, MW, 1>), grid, dim3(256), 0, st, p)
#define MVD_LAUNCH_U(L, DPB, MW) hipLaunchKernelGGL((warp_variance_kernel<L, WARP_ONLY, DPB, MW, 2>), grid, dim3(256), 0, st, p)
#define MVD_LAUNCH_V(L, DPB, MW) hipLaunchKernelGGL((warp_variance_kernel<L, WARP_ONLY, DPB, MW, 4>), grid, dim3(256), 0, st, p)
    if (p.exact_grid) {
        switch (lpp) {
            case 1: hipLaunchKernelGGL((warp_variance_kernel<1, WARP_ONLY, 4, 3, 0, true>), grid, dim3(256), 0, st, p); break;
            case 2: hipLaunchKernelGGL((warp_variance_kernel<2, WARP_ONLY, 4, 3, 0, true>), grid, dim3(256), 0, st, p); break;
            case 4: hipLaunchKernelGGL((warp_variance_kernel<4, WARP_ONLY, 4, 3, 0, true>), grid, dim3(256), 0, st, p); break;
            case 8: hipLaunchKernelGGL((warp_variance_kernel<8, WARP_ONLY, 4, 3, 0, true>), grid, dim3(256), 0, st, p); break;
            case 16: hipLaunchKernelGGL((warp_variance_kernel<16, WARP_ONLY, 4, 3, 0, true>), grid, dim3(256), 0, st, p); break;
        }
    } else if (lpp == 8) {
        switch (reuse * 1000 + dpb * 10 + minw) {
            case 2043: MVD_LAUNCH_U(8, 4, 3); break;  // the product kernel
#ifdef MVD_EXPERIMENTS
            case 18: MVD_LAUNCH(8, 1, 8); break;
            case 24: MVD_LAUNCH(8, 2, 4); break;
            case 28: MVD_LAUNCH(8, 2, 8); break;
            case 26: MVD_LAUNCH(8, 2, 6); break;
            case 42: MVD_LAUNCH(8, 4, 2); break;
            case 43: MVD_LAUNCH(8, 4, 3); break;
            case 44: MVD_LAUNCH(8, 4, 4); break;
            case 82: MVD_LAUNCH(8, 8, 2); break;
            case 1024: MVD_LAUNCH_R(8, 2, 4); break;
            case 1026: MVD_LAUNCH_R(8, 2, 6); break;
            case 1042: MVD_LAUNCH_R(8, 4, 2); break;
            case 1043: MVD_LAUNCH_R(8, 4, 3); break;
            case 1044: MVD_LAUNCH_R(8, 4, 4); break;
            case 1082: MVD_LAUNCH_R(8, 8, 2); break;
            case 2042: MVD_LAUNCH_U(8, 4, 2); break;
            case 2044: MVD_LAUNCH_U(8, 4, 4); break;
            case 4042: MVD_LAUNCH_V(8, 4, 2); break;
            case 4043: MVD_LAUNCH_V(8, 4, 3); break;
            case 4044: MVD_LAUNCH_V(8, 4, 4); break;
            case 83: MVD_LAUNCH(8, 8, 3); break;
#endif
            default:
                set_error("warp_variance: MVD_K3_CFG=%d,%d is not a compiled variant", dpb, minw);
                return MVD_ERR_INVALID_ARG;
        }
    } else {
        switch (lpp) {
            case 1: MVD_LAUNCH(1, 4, 3); break;
            case 2: MVD_LAUNCH(2, 4, 3); break;
            case 4: MVD_LAUNCH(4, 4, 3); break;
            case 16: MVD_LAUNCH(16, 4, 3); break;
        }
    }
#undef MVD_LAUNCH
#undef MVD_LAUNCH_R
#undef MVD_LAUNCH_U
#undef MVD_LAUNCH_V
    timing_end(st);
    return launch_status("warp_variance");
}

// max |x| over the FINITE values of n floats into *absmax (zeroed first): NaN and inf do not take part, so that one bad voxel
// does not set the scale of everything else.  A streaming read: 16 bytes per lane, grid-stride.
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
    __shared__ float wmax[4];
    const long long n4 = n / 4;
    float m = 0.f;
    auto take = [&m](const float4 v) {
        m = fmaxf(fmaxf(m, fmaxf(finite_abs_or_zero(v.x), finite_abs_or_zero(v.y))), fmaxf(finite_abs_or_zero(v.z), finite_abs_or_zero(v.w)));
    };
    // a contiguous segment per workgroup, eight 16-byte loads in flight per thread (with one load in flight the pass ran at 1.4 TB/s)
    const long long per_blk = ((n4 + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
    long long i = (long long)blockIdx.x * per_blk + threadIdx.x;
    const long long end = min(n4, ((long long)blockIdx.x + 1) * per_blk);
    for (; i + 7 * 256 < end; i += 8 * 256) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = x4[i + k * 256];
#pragma unroll
        for (int k = 0; k < 8; ++k) take(v[k]);
    }
    for (; i < end; i += 256) take(x4[i]);
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) m = fmaxf(m, finite_abs_or_zero(x[n4 * 4 + threadIdx.x]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    // ONE atomic per workgroup: atomics on a single address serialise at ~10 ns each
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        raise_absmax(reinterpret_cast<float*>(out), m);
    }
}

int absmax_launch(const float* x, long long n, float* absmax, hipStream_t st) {
    if (hipMemsetAsync(absmax, 0, sizeof(float), st) != hipSuccess) return launch_status("absmax: memset");
    if (n <= 0) return MVD_OK;
    // 16 loads of 16 bytes per thread, at most 8 workgroups per CU; a workgroup ends with at most one atomic
    const long long want = (n / 4 + 4095) / 4096;
    const unsigned nblk = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    hipLaunchKernelGGL(absmax_kernel, dim3(nblk), dim3(256), 0, st, x, n, reinterpret_cast<unsigned*>(absmax));
    return launch_status("absmax");
}

// shared by the two entry points: validates, lays out the workspace, repacks, composes, launches
static int run_warp(const float* key_feat, const float* const* src_feat, const float* const* src_proj,
                    const float* key_proj_inv, const float* depth_values, int B, int C, int D, int h, int w, int V,
                    float* out, int layout, void* workspace, size_t workspace_bytes, hipStream_t st, bool warp_only,
                    float* absmax = nullptr, bool* absmax_done = nullptr) {
    const char* who = warp_only ? "homo_warp" : "warp_variance";
    MVD_REQUIRE(B > 0 && D > 0 && h > 1 && w > 1, "%s: non-positive dimension (h, w must be >= 2)", who);
    MVD_REQUIRE(V >= 1 && V <= MVD_MAX_VIEWS, "%s: V=%d outside 1..%d", who, V, MVD_MAX_VIEWS);
    MVD_REQUIRE(C == 4 || C == 8 || C == 16 || C == 32 || C == 64, "%s: C=%d unsupported (need 4, 8, 16, 32 or 64)", who, C);
    MVD_REQUIRE((long long)(h + 3) * (w + 3) * C * 4 < 0x7fffffffLL && h < (1 << 23) && w < (1 << 23),
                "%s: one padded feature map of %dx%dx%d floats exceeds the 2 GiB buffer-offset range", who, h, w, C);
    // MVD_FEAT_NHWC_BORDER: the caller's maps already are zero-bordered channel-last copies (K6 writes them); only the
    // composed transforms need workspace
    const bool staged = (layout & MVD_FEAT_NHWC_BORDER) != 0;
    const size_t need = staged ? align_up((size_t)MVD_MAX_VIEWS * B * 12 * sizeof(float), 256)
                               : mvd_warp_variance_workspace_bytes(B, C, h, w, warp_only ? 0 : V);
    if (!workspace || workspace_bytes < need) {
        set_error("%s: workspace %zu B < required %zu B", who, workspace_bytes, need);
        return MVD_ERR_WORKSPACE;
    }
    const size_t slot = padded_slot_bytes(B, C, h, w);
    char* ws = (char*)workspace;
    WarpParams p{};
    p.M = (float*)ws;
    ws += align_up((size_t)MVD_MAX_VIEWS * B * 12 * sizeof(float), 256);
    int rc;
    if (!warp_only) {
        if (staged) {
            p.key = key_feat;
        } else {
            rc = repack_padded(key_feat, (float*)ws, B, C, h, w, st);
            if (rc) return rc;
            p.key = (float*)ws;
            ws += slot;
        }
    }
    for (int v = 0; v < V; ++v) {
        MVD_REQUIRE(src_feat[v] && src_proj[v], "%s: NULL view %d", who, v);
        if (staged) {
            p.src.p[v] = src_feat[v];
        } else {
            rc = repack_padded(src_feat[v], (float*)ws, B, C, h, w, st);
            if (rc) return rc;
            p.src.p[v] = (float*)ws;
            ws += slot;
        }
        p.proj.p[v] = src_proj[v];
    }
    hipLaunchKernelGGL(compose_transforms_kernel, dim3((unsigned)((V * B * 12 + 255) / 256)), dim3(256), 0, st, p.proj,
                       key_proj_inv, B, V, const_cast<float*>(p.M));
    rc = launch_status("compose_transforms");
    if (rc) return rc;
    p.key_proj_inv = key_proj_inv;
    p.depth = depth_values;
    p.out = out;
    p.B = B; p.D = D; p.h = h; p.w = w; p.V = V;
    p.layout = layout & 0xff;
    p.exact_grid = (layout & MVD_GRID_EXACT) ? 1 : 0;
    if (!warp_only && C == 32 && p.layout == MVD_LAYOUT_NDHWC) {
        int located_minw = 4, ko = 0;
        bool tile = warp_tile_supported(p, false);
        int tile_tw = 8, tile_win = 104, tile_nch = 8, tile_sets = 1;
#ifdef MVD_EXPERIMENTS
        if (const char* e = exp_env("MVD_K3_CFG")) {
            tile = false;
            if (e[0] == 'T') {  // "T<tw>,<win>,<nch>,<sets>": the LDS-footprint tile kernel (warp_variance_tile.hip)
                sscanf(e, "T%d,%d,%d,%d", &tile_tw, &tile_win, &tile_nch, &tile_sets);
                return launch_warp_tile(p, st, tile_tw, tile_win, tile_nch, tile_sets, false, p.exact_grid != 0);
            }
        }
#endif
        // product: LDS-staged footprints, 8x4 key tiles, 8-plane chunks (tools/bench_k3.py)
        if (tile) {
            if (absmax) {
                if (hipMemsetAsync(absmax, 0, sizeof(float), st) != hipSuccess) return launch_status("warp_variance: memset");
                p.absmax = absmax;
                *absmax_done = true;  // a by-product of the tile kernel; after any other kernel the caller runs a separate pass
            }
            return launch_warp_tile(p, st, tile_tw, tile_win, tile_nch, tile_sets, false, p.exact_grid != 0);
        }
        if (p.exact_grid) return launch_warp<false>(p, C, st);
#ifdef MVD_EXPERIMENTS
        if (const char* e = exp_env("MVD_K3_CFG")) {
            if (e[0] == 'L') { sscanf(e, "L%d,%d", &located_minw, &ko); if (located_minw == 4 && ko == 0) ko = -1; }  // "L3"/"L4": the located kernel; "L4,<ko>": knock-out
            else if (e[0] == 'M') {  // "M<minw>,<nsets>,<nch>": the marching form
                int mw = 5, ns = 2, nc = 4;
                sscanf(e, "M%d,%d,%d", &mw, &ns, &nc);
                return launch_warp_march(p, st, mw, ns, nc);
            } else located_minw = 0;                                    // any other selector: the round-1 kernels
        }
#endif
        // product: the marching form, 4 chunks per workgroup, two cells in flight, 4 waves per SIMD (tools/bench_k3.py)
        if (ko < 0) return launch_warp_located(p, st, 4, 0);
        if (located_minw == 4 && ko == 0 && (long long)p.h * p.w * 128 < 0x7fffffffLL && march_fits(p.V)) return launch_warp_march(p, st, 4, 2, 4);
        if (located_minw && (size_t)p.V * 4 * 32 * 20 > 64 * 1024) located_minw = 0;  // the located kernel: 2.5 KiB of LDS per view
        if (located_minw) return launch_warp_located(p, st, located_minw, ko);
#ifdef MVD_EXPERIMENTS
        // MVD_K3_CFG="lds,nd" selects the LDS-staged form (experiments library only)
        if (const char* e = exp_env("MVD_K3_CFG")) {
            if (e[0] == 'l') {
                int nd = 4;
                sscanf(e, "lds,%d", &nd);
                return launch_warp_lds(p, st, nd);
            } else if (e[0] == 'q') {
                int mw = 4;
                sscanf(e, "q8,%d", &mw);
                return launch_warp_q8(p, st, mw);
            } else if (e[0] == 'w') {
                int nd = 8;
                sscanf(e, "wave,%d", &nd);
                return launch_warp_wave(p, st, nd);
            }
        }
#endif
    }
    return warp_only ? launch_warp<true>(p, C, st) : launch_warp<false>(p, C, st);
}

// fp16-feature variant (BASELINE configs[3]): features arrive as fp16 zero-bordered channel-last maps, the volume
// leaves as fp16 channel-last; everything in between is the fp32 arithmetic of the marching kernel
static int run_warp_f16(const void* key_feat, const void* const* src_feat, const float* const* src_proj,
                        const float* key_proj_inv, const float* depth_values, int B, int D, int h, int w, int V, void* out,
                        void* workspace, size_t workspace_bytes, hipStream_t st) {
    const char* who = "warp_variance_f16";
    MVD_REQUIRE(B > 0 && D > 0 && h > 1 && w > 1, "%s: non-positive dimension (h, w must be >= 2)", who);
    MVD_REQUIRE(V >= 1 && V <= MVD_MAX_VIEWS, "%s: V=%d outside 1..%d", who, V, MVD_MAX_VIEWS);
    MVD_REQUIRE((long long)(h + 3) * (w + 3) * 64 < 0x7fffffffLL && h < (1 << 23) && w < (1 << 23),
                "%s: one padded feature map of %dx%dx32 halves exceeds the 2 GiB buffer-offset range", who, h, w);
    const size_t need = align_up((size_t)MVD_MAX_VIEWS * B * 12 * sizeof(float), 256);
    if (!workspace || workspace_bytes < need) {
        set_error("%s: workspace %zu B < required %zu B", who, workspace_bytes, need);
        return MVD_ERR_WORKSPACE;
    }
    WarpParams p{};
    p.M = (float*)workspace;
    p.key = (const float*)key_feat;
    for (int v = 0; v < V; ++v) {
        MVD_REQUIRE(src_feat[v] && src_proj[v], "%s: NULL view %d", who, v);
        p.src.p[v] = (const float*)src_feat[v];
        p.proj.p[v] = src_proj[v];
    }
    hipLaunchKernelGGL(compose_transforms_kernel, dim3((unsigned)((V * B * 12 + 255) / 256)), dim3(256), 0, st, p.proj,
                       key_proj_inv, B, V, const_cast<float*>(p.M));
    int rc = launch_status("compose_transforms");
    if (rc) return rc;
    p.key_proj_inv = key_proj_inv;
    p.depth = depth_values;
    p.out = (float*)out;
    p.B = B; p.D = D; p.h = h; p.w = w; p.V = V;
    p.layout = MVD_LAYOUT_NDHWC;
#ifdef MVD_EXPERIMENTS
    if (const char* e = exp_env("MVD_K3_CFG")) {
        if (e[0] == 'T') {
            int tw = 8, win = 128, nch = 16, sets = 2;
            sscanf(e, "T%d,%d,%d,%d", &tw, &win, &nch, &sets);
            return launch_warp_tile(p, st, tw, win, nch, sets, true, false);
        }
        if (e[0] == 'M') return launch_warp_march(p, st, 4, 2, 4, true);
    }
#endif
    if (warp_tile_supported(p, true)) return launch_warp_tile(p, st, 8, 128, 8, 1, true, false);
    if (!march_fits(V)) {
        set_error("%s: maps of %dx%d with V=%d: neither the tile kernel (h, w < 65533) nor the marching kernel (V <= 12) applies", who, h, w, V);
        return MVD_ERR_INVALID_ARG;
    }
    return launch_warp_march(p, st, 4, 2, 4, true);
}

template <bool TO_HALF>
__global__ void __launch_bounds__(256) convert_kernel(const void* __restrict__ src, void* __restrict__ dst, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // 4 elements per thread
    if (i >= n4) return;
    if constexpr (TO_HALF) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        const f16x2 lo = {(_Float16)v.x, (_Float16)v.y}, hi = {(_Float16)v.z, (_Float16)v.w};
        reinterpret_cast<u32x2*>(dst)[i] = u32x2{as_u32(lo), as_u32(hi)};
    } else {
        const u32x2 v = reinterpret_cast<const u32x2*>(src)[i];
        const f16x2 lo = as_h2(v.x), hi = as_h2(v.y);
        reinterpret_cast<float4*>(dst)[i] = make_float4((float)lo.x, (float)lo.y, (float)hi.x, (float)hi.y);
    }
}

template <bool TO_HALF>
static int convert(const void* src, void* dst, long long n, hipStream_t st) {
    MVD_REQUIRE(src && dst && n > 0 && n % 4 == 0, "convert: NULL argument or element count not a positive multiple of 4");
    const long long n4 = n / 4, nblk = (n4 + 255) / 256;
    MVD_REQUIRE(nblk <= 0x7fffffffLL, "convert: too many elements");
    hipLaunchKernelGGL(convert_kernel<TO_HALF>, dim3((unsigned)nblk), dim3(256), 0, st, src, dst, n4);
    return launch_status("convert");
}

}  // namespace mvd

extern "C" {

size_t mvd_warp_variance_f16_workspace_bytes(int B) {
    return B > 0 ? mvd::align_up((size_t)MVD_MAX_VIEWS * B * 12 * sizeof(float), 256) : 0;
}

int mvd_warp_variance_f16(const void* key_feat, const void* const* src_feat, const float* const* src_proj,
                          const float* key_proj_inv, const float* depth_values, int B, int D, int h, int w, int V,
                          void* var_out, void* workspace, size_t workspace_bytes, mvd_stream_t stream) {
    MVD_REQUIRE(key_feat && src_feat && src_proj && key_proj_inv && depth_values && var_out, "warp_variance_f16: NULL argument");
    return mvd::run_warp_f16(key_feat, src_feat, src_proj, key_proj_inv, depth_values, B, D, h, w, V, var_out, workspace,
                             workspace_bytes, (hipStream_t)stream);
}

int mvd_convert_f32_to_f16(const float* src, void* dst, long long n, mvd_stream_t stream) {
    return mvd::convert<true>(src, dst, n, (hipStream_t)stream);
}
int mvd_convert_f16_to_f32(const void* src, float* dst, long long n, mvd_stream_t stream) {
    return mvd::convert<false>(src, dst, n, (hipStream_t)stream);
}

size_t mvd_warp_variance_workspace_bytes(int B, int C, int h, int w, int V) {
    if (B <= 0 || C <= 0 || h <= 0 || w <= 0 || V < 0) return 0;
    // transforms + (V + 1) zero-bordered channel-last feature copies (V = 0: homo_warp, one copy)
    return mvd::align_up((size_t)MVD_MAX_VIEWS * B * 12 * sizeof(float), 256) +
           (size_t)(V + 1) * mvd::padded_slot_bytes(B, C, h, w);
}

int mvd_warp_variance_f32(const float* key_feat, const float* const* src_feat, const float* const* src_proj,
                          const float* key_proj_inv, const float* depth_values, int B, int C, int D, int h, int w,
                          int V, float* var_out, int out_layout, void* workspace, size_t workspace_bytes,
                          mvd_stream_t stream) {
    MVD_REQUIRE(key_feat && src_feat && src_proj && key_proj_inv && depth_values && var_out,
                "warp_variance: NULL argument");
    MVD_REQUIRE((out_layout & 0xff) == MVD_LAYOUT_NCDHW || (out_layout & 0xff) == MVD_LAYOUT_NDHWC, "warp_variance: bad layout");
    return mvd::run_warp(key_feat, src_feat, src_proj, key_proj_inv, depth_values, B, C, D, h, w, V, var_out, out_layout,
                         workspace, workspace_bytes, (hipStream_t)stream, false);
}

int mvd_warp_variance_absmax_f32(const float* key_feat, const float* const* src_feat, const float* const* src_proj,
                                 const float* key_proj_inv, const float* depth_values, int B, int C, int D, int h, int w,
                                 int V, float* var_out, float* absmax_out, int out_layout, void* workspace, size_t workspace_bytes,
                                 mvd_stream_t stream) {
    MVD_REQUIRE(key_feat && src_feat && src_proj && key_proj_inv && depth_values && var_out && absmax_out,
                "warp_variance_absmax: NULL argument");
    MVD_REQUIRE((out_layout & 0xff) == MVD_LAYOUT_NCDHW || (out_layout & 0xff) == MVD_LAYOUT_NDHWC, "warp_variance: bad layout");
    bool done = false;
    int rc = mvd::run_warp(key_feat, src_feat, src_proj, key_proj_inv, depth_values, B, C, D, h, w, V, var_out, out_layout,
                           workspace, workspace_bytes, (hipStream_t)stream, false, absmax_out, &done);
    if (rc == MVD_OK && !done) rc = mvd::absmax_launch(var_out, (long long)B * C * D * h * w, absmax_out, (hipStream_t)stream);
    return rc;
}

int mvd_absmax_f32(const float* x, long long n, float* absmax, mvd_stream_t stream) {
    MVD_REQUIRE(x && absmax && n >= 0, "absmax: NULL argument or negative count");
    return mvd::absmax_launch(x, n, absmax, (hipStream_t)stream);
}

int mvd_homo_warp_f32(const float* src_feat, const float* src_proj, const float* key_proj_inv,
                      const float* depth_values, int B, int C, int D, int h, int w, float* warped_out,
                      void* workspace, size_t workspace_bytes, mvd_stream_t stream) {
    MVD_REQUIRE(src_feat && src_proj && key_proj_inv && depth_values && warped_out, "homo_warp: NULL argument");
    const float* srcs[1] = {src_feat};
    const float* projs[1] = {src_proj};
    return mvd::run_warp(nullptr, srcs, projs, key_proj_inv, depth_values, B, C, D, h, w, 1, warped_out,
                         MVD_LAYOUT_NCDHW, workspace, workspace_bytes, (hipStream_t)stream, true);
}
}

