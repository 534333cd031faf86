// Input resize of the model adapters on the device.
// Replaces ResizeInputs / UpscaleInputsToNextMultipleOf (rmvd/data/transforms.py:40-98), i.e.
// skimage.transform.resize(image, (..., ht, wd), order=1) for the only case the adapters produce: UPSCALING to the next
// multiple of 64 / 32 (robust_mvd.py:104-113, mvsnet.py:178).  For that case skimage applies no anti-aliasing filter,
// keeps float32 and delegates to scipy.ndimage.zoom(order=1, mode='mirror', grid_mode=True), whose arithmetic this
// kernel follows operation for operation in float64:
//   zoom = n_in / n_out;  c = (o + 0.5) * zoom - 0.5;  mirror c into [0, n_in - 1];  i0 = floor(c), w1 = c - i0,
//   w0 = 1 - w1, i1 = i0 + 1 (mirrored);  t = ((a00*wy0)*wx0 + (a01*wy0)*wx1) + (a10*wy1)*wx0 + (a11*wy1)*wx1.
// HBM-bound, trivially small (5 images of 3.7 MB per frame): one thread per output pixel.
#include "mvd_common.h"

namespace mvd {

struct AxisTap {
    int i0, i1;
    double w0, w1;
};

__device__ __forceinline__ AxisTap axis_tap(int o, int n_in, int n_out) {
    AxisTap t;
    if (n_in <= 1) {
        t.i0 = t.i1 = 0; t.w0 = 1.0; t.w1 = 0.0;
        return t;
    }
    const double zoom = (double)n_in / (double)n_out;
    double c = ((double)o + 0.5) * zoom - 0.5;
    const double last = (double)(n_in - 1), sz2 = 2.0 * last;
    if (c < 0.0) c = -c;          // upscaling: |c| < 0.5, one reflection is enough
    if (c > last) c = sz2 - c;
    const double st = floor(c);
    t.w1 = c - st;
    t.w0 = 1.0 - t.w1;
    t.i0 = (int)st;
    t.i1 = t.i0 + 1;
    if (t.i1 > n_in - 1) t.i1 = 2 * (n_in - 1) - t.i1;
    return t;
}

__global__ void __launch_bounds__(256) resize_order1_kernel(const float* __restrict__ src, float* __restrict__ dst, int hi, int wi,
                                                            int ho, int wo) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    const size_t plane = blockIdx.z;
    if (x >= wo) return;
    const AxisTap ty = axis_tap(y, hi, ho), tx = axis_tap(x, wi, wo);
    const float* __restrict__ s = src + plane * (size_t)hi * wi;
    const double a00 = s[(size_t)ty.i0 * wi + tx.i0], a01 = s[(size_t)ty.i0 * wi + tx.i1];
    const double a10 = s[(size_t)ty.i1 * wi + tx.i0], a11 = s[(size_t)ty.i1 * wi + tx.i1];
    double t = (a00 * ty.w0) * tx.w0;  // -ffp-contract=off: one rounding per operation, like the reference's C loop
    t = t + (a01 * ty.w0) * tx.w1;
    t = t + (a10 * ty.w1) * tx.w0;
    t = t + (a11 * ty.w1) * tx.w1;
    dst[(plane * ho + y) * (size_t)wo + x] = (float)t;
}

}  // namespace mvd

extern "C" int mvd_resize_order1_f32(const float* src, float* dst, long long planes, int hi, int wi, int ho, int wo,
                                     mvd_stream_t stream) {
    MVD_REQUIRE(src && dst, "resize_order1: NULL argument");
    MVD_REQUIRE(planes > 0 && planes <= 65535 && hi > 0 && wi > 0, "resize_order1: bad dimensions (planes 1..65535)");
    MVD_REQUIRE(ho >= hi && wo >= wi && ho <= 65535,
                "resize_order1: only upscaling is built (%dx%d -> %dx%d): downscaling takes skimage's anti-aliasing filter", hi,
                wi, ho, wo);
    const dim3 grid((unsigned)((wo + 255) / 256), (unsigned)ho, (unsigned)planes);
    hipLaunchKernelGGL(mvd::resize_order1_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, hi, wi, ho, wo);
    return mvd::launch_status("resize_order1");
}
