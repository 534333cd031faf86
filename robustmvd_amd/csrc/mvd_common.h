// Shared host/device helpers of libmvd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "mvd.h"

namespace mvd {

void set_error(const char* fmt, ...);
// measurement hook (mvd_arm_kernel_timing): record the armed events around a main-kernel launch
void timing_begin(hipStream_t st);
void timing_end(hipStream_t st);
// max |x| over the finite values of n floats into *absmax (device, one float; zeroed first): warp_variance.hip
int absmax_launch(const float* x, long long n, float* absmax, hipStream_t st);
// Raise *slot (a non-negative float kept as its bit pattern) to m.  Atomics on one address serialise (~10 ns each), and after the
// first few workgroups of a launch most arrive with a smaller value: those find that out with a plain load and leave.
__device__ __forceinline__ void raise_absmax(float* slot, float m) {
    if (m > 0.f && m > __builtin_nontemporal_load(slot)) atomicMax(reinterpret_cast<unsigned*>(slot), __float_as_uint(m));
}
// The same with a first check against a value of the slot read EARLY (at kernel start) by the caller: a short workgroup that reads the
// slot at its end holds its CU slot for one more memory latency (K6's conv1: +20 us over 17,000 workgroups), and most workgroups
// are below what the slot held when they started.  Only those above it look again (the early value alone is not enough: while the
// maximum is still growing, every workgroup in flight would fire its atomic at the one address: conv1 413 instead of 140 us).
__device__ __forceinline__ float absmax_seen(const float* slot) { return slot ? __builtin_nontemporal_load(slot) : 0.f; }
__device__ __forceinline__ void raise_absmax_seen(float* slot, float m, float seen) {
    if (m > seen) raise_absmax(slot, m);
}
__device__ __forceinline__ float finite_abs_or_zero(float v) {
    const float a = fabsf(v);
    return a <= 3.402823466e38f ? a : 0.f;  // false for inf and NaN
}

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MVD_ERR_LAUNCH;
    }
    return MVD_OK;
}

#define MVD_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            mvd::set_error(__VA_ARGS__); \
            return MVD_ERR_INVALID_ARG; \
        }                               \
    } while (0)

// per-view device pointers, passed to kernels by value (kernarg segment)
struct ViewPtrs {
    const float* p[MVD_MAX_VIEWS];
};
struct ViewOutPtrs {
    float* p[MVD_MAX_VIEWS];
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Kernel-variant selectors for experiments.  The PRODUCT library (default build) has fixed dispatch and never reads
// the process environment (include/mvd.h: no global mutable state): exp_env() is a constant nullptr there and every
// `if (exp_env(...))` branch folds away.  Only the -DMVD_EXPERIMENTS build (`make exp` -> robustmvd_amd/lib_exp/
// libmvd_hip_exp.so, used by tools/ and the variants test) looks at MVD_K3_CFG / MVD_K4_*.
#ifdef MVD_EXPERIMENTS
const char* exp_env(const char* name);
#else
constexpr const char* exp_env(const char*) { return nullptr; }
#endif

// ---- device: exactly rounded fp32 steps for the sampling-grid arithmetic -----------------------
// The grids decide which taps are in bounds (a 0/1 mask in Path A), so they follow the reference's
// operation order with one rounding per operation; the library is built with -ffp-contract=off and
// the hot accumulation loops ask for FMAs explicitly (fmaf).
__device__ __forceinline__ float unnormalize_coord(float g, float size) {
    // ATen grid_sampler_unnormalize, align_corners=False: ((g + 1) * size - 1) / 2
    return ((g + 1.0f) * size - 1.0f) / 2.0f;
}

struct Taps {
    int off[4];  // pixel index (y*ws + x) of nw, ne, sw, se; 0 when out of bounds
    float w[4];  // bilinear weight; 0 when out of bounds
    float inb;   // sum of in-bounds weights
};

// bilinear taps with zero padding at un-normalised source index (ix, iy); NaN/inf coordinates fall
// out of bounds on every tap (all comparisons false), like ATen's CPU kernel.
__device__ __forceinline__ Taps bilinear_taps(float ix, float iy, int hs, int ws) {
    Taps t;
    const float x0 = floorf(ix), y0 = floorf(iy);
    const float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    const float wx1 = ix - x0, wx0 = x1 - ix, wy1 = iy - y0, wy0 = y1 - iy;
    const float xs[4] = {x0, x1, x0, x1};
    const float ys[4] = {y0, y0, y1, y1};
    const float wt[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
    const float xmax = (float)(ws - 1), ymax = (float)(hs - 1);
    t.inb = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool in = (xs[k] >= 0.0f) && (xs[k] <= xmax) && (ys[k] >= 0.0f) && (ys[k] <= ymax);
        t.off[k] = in ? ((int)ys[k] * ws + (int)xs[k]) : 0;
        t.w[k] = in ? wt[k] : 0.0f;
        t.inb += t.w[k];
    }
    return t;
}

}  // namespace mvd
