// Shared by warp_variance.hip (the product kernels) and warp_variance_exp.hip (experimental variants, compiled only
// into the -DMVD_EXPERIMENTS library).
#pragma once
#include "mvd_common.h"

namespace mvd {

struct WarpParams {
    ViewPtrs src;           // V x (B,h+3,w+3,C) zero-bordered channel-last source features
    ViewPtrs proj;          // V x (B,4,4) source projection matrices
    const float* key;       // (B,h+3,w+3,C) zero-bordered channel-last key features (unused when WARP_ONLY)
    const float* M;         // (V,B,12) composed transforms
    const float* key_proj_inv;  // (B,4,4)
    const float* depth;     // (B,D)
    float* out;
    int B, D, h, w, V;
    int layout;             // MVD_LAYOUT_*
    int tiles_x, tiles_y, tiles_per_xcd;  // filled by the launchers
    int exact_grid;  // 1: sampling positions follow the reference's operation chain rounding for rounding
    float* absmax;   // optional (device, one float, zeroed by the launcher): receives max |out| (tile kernel; fp32 volume)
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// One plane of one source view: bilinear blend of the cell's 4 taps (4 channels per lane), then the running sum and sum
// of squares of mvsnet.py:131-134.
__device__ __forceinline__ void accumulate_cell(float4& a1, float4& a2, const float (&w)[4], const u32x4 (&t)[4]) {
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        acc.x = fmaf(__uint_as_float(t[k].x), w[k], acc.x);
        acc.y = fmaf(__uint_as_float(t[k].y), w[k], acc.y);
        acc.z = fmaf(__uint_as_float(t[k].z), w[k], acc.z);
        acc.w = fmaf(__uint_as_float(t[k].w), w[k], acc.w);
    }
    a1.x += acc.x; a1.y += acc.y; a1.z += acc.z; a1.w += acc.w;
    a2.x = fmaf(acc.x, acc.x, a2.x); a2.y = fmaf(acc.y, acc.y, a2.y);
    a2.z = fmaf(acc.z, acc.z, a2.z); a2.w = fmaf(acc.w, acc.w, a2.w);
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// by VALUE on purpose: __builtin_bit_cast applied directly to a vector component (t.y) reads the vector's first dword
__device__ __forceinline__ f16x2 as_h2(unsigned v) { return __builtin_bit_cast(f16x2, v); }
__device__ __forceinline__ unsigned as_u32(f16x2 v) { return __builtin_bit_cast(unsigned, v); }

// fp16 taps (4 channels = 8 bytes per lane), fp32 weights and accumulation: fmaf((float)half, w, acc) is one
// v_fma_mix_f32 (exact f16 -> f32 conversion inside the FMA), so the result equals the fp32 kernel's on the same
// (fp16-representable) feature values, operation for operation
__device__ __forceinline__ void accumulate_cell(float4& a1, float4& a2, const float (&w)[4], const u32x2 (&t)[4]) {
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f16x2 lo = as_h2(t[k].x), hi = as_h2(t[k].y);
        acc.x = fmaf((float)lo.x, w[k], acc.x);
        acc.y = fmaf((float)lo.y, w[k], acc.y);
        acc.z = fmaf((float)hi.x, w[k], acc.z);
        acc.w = fmaf((float)hi.y, w[k], acc.w);
    }
    a1.x += acc.x; a1.y += acc.y; a1.z += acc.z; a1.w += acc.w;
    a2.x = fmaf(acc.x, acc.x, a2.x); a2.y = fmaf(acc.y, acc.y, a2.y);
    a2.z = fmaf(acc.z, acc.z, a2.z); a2.w = fmaf(acc.w, acc.w, a2.w);
}

// (the b64 builtins traffic in GCC-style vectors; an implicit conversion to an ext_vector_type silently narrows the load
// to one dword with this compiler, so the lanes are moved explicitly)
typedef unsigned int u32x2n __attribute__((vector_size(8)));
__device__ __forceinline__ u32x2 load_b64(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    const u32x2n r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
    return u32x2{r[0], r[1]};
}
__device__ __forceinline__ void store_b64(u32x2 v, __amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    const u32x2n r = {v.x, v.y};
    __builtin_amdgcn_raw_buffer_store_b64(r, rsrc, voff, 0, 0);
}

// LDS-footprint tile kernel (warp_variance_tile.hip): C = 32, channel-last volume
bool warp_tile_supported(const WarpParams& p, bool f16);
int launch_warp_tile(const WarpParams& p0, hipStream_t st, int tw, int win, int nch, int sets, bool f16, bool exact);

#ifdef MVD_EXPERIMENTS
// experimental launchers (warp_variance_exp.hip); each returns an mvd_status
int launch_warp_q8(const WarpParams& p0, hipStream_t st, int minw);
int launch_warp_wave(const WarpParams& p0, hipStream_t st, int nd);
int launch_warp_lds(const WarpParams& p0, hipStream_t st, int nd);
#endif

}  // namespace mvd
