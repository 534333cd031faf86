// Version / error reporting / layout helpers of libmvd_hip.so.
#include "mvd_common.h"

namespace mvd {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

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
}  // namespace mvd

extern "C" {
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
