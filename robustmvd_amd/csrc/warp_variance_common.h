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

#ifdef MVD_EXPERIMENTS
// experimental launchers (warp_variance_exp.hip); each returns an mvd_status
int launch_warp_q8(const WarpParams& p0, hipStream_t st, int minw);
int launch_warp_wave(const WarpParams& p0, hipStream_t st, int nd);
int launch_warp_lds(const WarpParams& p0, hipStream_t st, int nd);
#endif

}  // namespace mvd
