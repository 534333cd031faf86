// Version / error reporting / layout helpers of libmvd_hip.so.
#include "mvd_common.h"
#include <stdlib.h>

namespace mvd {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#ifdef MVD_EXPERIMENTS
const char* exp_env(const char* name) { return getenv(name); }  // experiments library only (mvd_common.h)
#endif

static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
void timing_begin(hipStream_t st) {
    if (g_ev_start) (void)hipEventRecord(g_ev_start, st);
}
void timing_end(hipStream_t st) {
    if (g_ev_stop) (void)hipEventRecord(g_ev_stop, st);
    g_ev_start = g_ev_stop = nullptr;
}

// (N, C, HW) <-> (N, HW, C) through a 32x33 LDS tile: both sides move 128-B rows.
__global__ void __launch_bounds__(256) transpose_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                        long long rows, long long cols) {
    // src: (N, rows, cols) -> dst: (N, cols, rows)
    __shared__ float tile[32][33];
    const long long ctiles = (cols + 31) / 32, rtiles = (rows + 31) / 32;
    long long bid = blockIdx.x;
    const long long ct = bid % ctiles;
    bid /= ctiles;
    const long long rt = bid % rtiles;
    const int n = (int)(bid / rtiles);
    const long long c0 = ct * 32;
    const long long r0 = rt * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const float* s = src + (long long)n * rows * cols;
    float* d = dst + (long long)n * rows * cols;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long r = r0 + ty + 8 * k;
        const long long c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 8 * k][tx] = s[(long long)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long c = c0 + ty + 8 * k;
        const long long r = r0 + tx;
        if (r < rows && c < cols) d[c * rows + r] = tile[tx][ty + 8 * k];
    }
}

int transpose_launch(const float* src, float* dst, int N, long long rows, long long cols, hipStream_t st) {
    const long long nblk = ((cols + 31) / 32) * ((rows + 31) / 32) * N;
    if (nblk > 0x7fffffffLL) {
        set_error("transpose: %lld tiles exceed the grid limit", nblk);
        return MVD_ERR_INVALID_ARG;
    }
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)nblk), dim3(256), 0, st, src, dst, rows, cols);
    return launch_status("transpose");
}
// x[n][c][i] = leaky_relu(x[n][c][i] + bias[c]) in place: the epilogue of the DispNet convolutions around the Path-A
// sweep (rmvd/models/blocks/dispnet_*.py: Conv2d(bias=True) + LeakyReLU(0.2)), which torch runs as two extra passes
template <bool VEC>
__global__ void __launch_bounds__(256) bias_leaky_relu_kernel(float* __restrict__ x, const float* __restrict__ bias, int C,
                                                              long long HW, float slope) {
    const long long row = blockIdx.y;  // n * C + c
    const float b = bias[row % C];
    float* __restrict__ xr = x + row * HW;
    if constexpr (VEC) {
        float4* __restrict__ x4 = reinterpret_cast<float4*>(xr);
        const long long n4 = HW / 4;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            float4 v = x4[i];
            v.x += b; v.y += b; v.z += b; v.w += b;
            v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
            v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
            x4[i] = v;
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long long)gridDim.x * 256) {
            const float v = xr[i] + b;
            xr[i] = v > 0.f ? v : v * slope;
        }
    }
}

// Prediction head of the DispNet decoder (rmvd/models/blocks/dispnet_decoder.py:17-22,126-138 with ReLUAndSigmoid,
// blocks/utils.py:30-41, min = -10, max = 10): from the 2-channel output x of a pred_k convolution
//   pred[:,0] = relu(x0)                                   inverse depth
//   pred[:,1] = sigmoid(x1 * (4/20)) * 20 + (-10)          log b
//   ent       = log(2 * exp(pred[:,1]) + 1e-4) + 1         entropy of the Laplace distribution
// in one pass: torch runs this as ~10 elementwise launches per head, 6 heads per frame.
__global__ void __launch_bounds__(256) dispnet_head_kernel(const float* __restrict__ x, float* __restrict__ pred,
                                                           float* __restrict__ ent, long long HW) {
    const long long n = blockIdx.y;
    const float* __restrict__ x0 = x + n * 2 * HW;
    float* __restrict__ p0 = pred + n * 2 * HW;
    float* __restrict__ e0 = ent + n * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long long)gridDim.x * 256) {
        const float a = x0[i], bq = x0[HW + i];
        const float sg = 1.0f / (1.0f + expf(-(bq * 0.2f)));
        const float lb = sg * 20.0f + -10.0f;
        p0[i] = fmaxf(a, 0.0f);
        p0[HW + i] = lb;
        e0[i] = logf(2.0f * expf(lb) + 1e-4f) + 1.0f;
    }
}
}  // namespace mvd

namespace mvd {
// measurement aid (bench.py): a pure streaming-store pass, 16 bytes per lane, 16 KiB contiguous per workgroup and step
__global__ void __launch_bounds__(256) stream_fill_kernel(float4* __restrict__ out, long long n4, float seed) {
    const float v = seed + threadIdx.x;
    const float4 r = make_float4(v, v + 1, v + 2, v + 3);
    for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (long long)gridDim.x * 1024) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k * 256 < n4) out[i + k * 256] = r;
    }
}
}  // namespace mvd

extern "C" {
int mvd_stream_fill_f32(float* dst, long long n, float value, mvd_stream_t stream) {
    MVD_REQUIRE(dst && n > 0 && n % 4 == 0 && ((size_t)dst % 16) == 0, "stream_fill: NULL / misaligned destination or count not a multiple of 4");
    const long long n4 = n / 4, want = (n4 + 1023) / 1024;
    hipLaunchKernelGGL(mvd::stream_fill_kernel, dim3((unsigned)(want > 16384 ? 16384 : want)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float4*>(dst), n4, value);
    return mvd::launch_status("stream_fill");
}
int mvd_dispnet_head_f32(const float* x, float* pred, float* ent, int N, long long HW, mvd_stream_t stream) {
    MVD_REQUIRE(x && pred && ent && N > 0 && N <= 65535 && HW > 0, "dispnet_head: bad argument");
    const unsigned gx = (unsigned)((HW + 255) / 256 > 1024 ? 1024 : (HW + 255) / 256);
    hipLaunchKernelGGL(mvd::dispnet_head_kernel, dim3(gx, (unsigned)N), dim3(256), 0, (hipStream_t)stream, x, pred, ent, HW);
    return mvd::launch_status("dispnet_head");
}
int mvd_bias_leaky_relu_f32(float* x, const float* bias, int N, int C, long long HW, float slope, mvd_stream_t stream) {
    MVD_REQUIRE(x && bias && N > 0 && C > 0 && HW > 0, "bias_leaky_relu: bad argument");
    MVD_REQUIRE((long long)N * C <= 65535, "bias_leaky_relu: N*C = %lld exceeds the grid limit", (long long)N * C);
    const bool vec = HW % 4 == 0 && ((size_t)x % 16) == 0;
    const long long per_row = vec ? HW / 4 : HW;
    const unsigned gx = (unsigned)((per_row + 255) / 256 > 256 ? 256 : (per_row + 255) / 256);
    const dim3 grid(gx, (unsigned)((long long)N * C));
    if (vec) hipLaunchKernelGGL(mvd::bias_leaky_relu_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, bias, C, HW, slope);
    else hipLaunchKernelGGL(mvd::bias_leaky_relu_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, bias, C, HW, slope);
    return mvd::launch_status("bias_leaky_relu");
}
int mvd_version(void) { return MVD_VERSION; }
const char* mvd_last_error(void) { return mvd::g_err; }
int mvd_arm_kernel_timing(void* start_event, void* stop_event) {
    mvd::g_ev_start = (hipEvent_t)start_event;
    mvd::g_ev_stop = (hipEvent_t)stop_event;
    return MVD_OK;
}

int mvd_nchw_to_nhwc_f32(const float* src, float* dst, int N, int C, long long HW, mvd_stream_t stream) {
    MVD_REQUIRE(src && dst && N > 0 && C > 0 && HW > 0, "nchw_to_nhwc: bad argument");
    return mvd::transpose_launch(src, dst, N, C, HW, (hipStream_t)stream);
}
int mvd_nhwc_to_nchw_f32(const float* src, float* dst, int N, int C, long long HW, mvd_stream_t stream) {
    MVD_REQUIRE(src && dst && N > 0 && C > 0 && HW > 0, "nhwc_to_nchw: bad argument");
    return mvd::transpose_launch(src, dst, N, HW, C, (hipStream_t)stream);  // src (N, HW, C): rows = HW, cols = C
}
}
