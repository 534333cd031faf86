// K5 — softmax over the depth axis + soft-argmin depth + 4-bin confidence (Path B).
// Replaces F.softmax + depth_regression + avg_pool3d/gather of MVSNet.forward
// (rmvd/models/mvsnet.py:139-160, rmvd/models/blocks/utils.py:271-274).
// Lanes along x: every load of a depth plane row is a coalesced 256-B segment.  One sweep over D (chunked online softmax, D split
// over the 4 waves of a workgroup) plus a 4-plane window re-read for the confidence; the cost volume is D*h*w*4 B (57 MB at the
// headline shape).
#include "mvd_common.h"

namespace mvd {

// A workgroup = 64 pixels x 4 waves; wave k sweeps depth planes [k Dq, (k+1) Dq) (Dq = D / 4 rounded up to a multiple of 8) with the
// chunked online softmax, the four partial results are merged through LDS in the fixed order k = 0 .. 3.  (One lane per pixel over
// all D planes left 216 workgroups of latency-bound lanes on 256 CUs at the headline shape: 45 us for a 57 MB read.)
__global__ void __launch_bounds__(256) softmax_regress_kernel(const float* __restrict__ cost,
                                                              const float* __restrict__ depth_values, int D,
                                                              long long hw, float* __restrict__ depth_out,
                                                              float* __restrict__ conf_out) {
    __shared__ float part[4][4][64];  // [wave][m, se, sd, si][lane]
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long pix = (long long)blockIdx.x * 64 + lane;
    const bool live = pix < hw;
    const float* c = cost + (long long)b * D * hw + (live ? pix : 0);
    const float* dv = depth_values + (long long)b * D;

    constexpr int CH = 8;
    const int Dq = ((D + 3) / 4 + CH - 1) / CH * CH;
    const int dbeg = wv * Dq, dend = min(D, dbeg + Dq);
    float m = -INFINITY, se = 0.f, sd = 0.f, si = 0.f;
    for (int d0 = dbeg; d0 < dend; d0 += CH) {
        float v[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) v[k] = d0 + k < dend ? c[(long long)(d0 + k) * hw] : -INFINITY;
        float cm = v[0];
#pragma unroll
        for (int k = 1; k < CH; ++k) cm = fmaxf(cm, v[k]);
        if (cm > m) {
            const float r = expf(m - cm);  // exp(-inf) = 0 on the first chunk
            se *= r; sd *= r; si *= r;
            m = cm;
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (d0 + k < dend) {
                const float e = expf(v[k] - m);
                se += e;
                sd = fmaf(e, dv[d0 + k], sd);
                si = fmaf(e, (float)(d0 + k), si);
            }
        }
    }
    part[wv][0][lane] = m; part[wv][1][lane] = se; part[wv][2][lane] = sd; part[wv][3][lane] = si;
    __syncthreads();
    if (wv != 0 || !live) return;
    // merge: common max, partial sums rescaled to it (a slice without planes has m = -inf and sums 0: exp(-inf) * 0 = 0)
    float M = part[0][0][lane];
#pragma unroll
    for (int k = 1; k < 4; ++k) M = fmaxf(M, part[k][0][lane]);
    se = 0.f; sd = 0.f; si = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float mk = part[k][0][lane];
        const float r = mk == -INFINITY ? 0.f : expf(mk - M);
        se = fmaf(part[k][1][lane], r, se);
        sd = fmaf(part[k][2][lane], r, sd);
        si = fmaf(part[k][3][lane], r, si);
    }
    // depth = sum_d p_d depth_d, expected index = sum_d p_d d (mvsnet.py:140-141,151-154) with p_d = e_d / se
    depth_out[(long long)b * hw + pix] = sd / se;
    if (conf_out) {
        const int idx = (int)(si / se);  // .long(): truncation (mvsnet.py:154)
        float conf = 0.f;
#pragma unroll
        for (int j = -1; j <= 2; ++j) {
            const int dd = idx + j;
            if (dd >= 0 && dd < D) conf += expf(c[(long long)dd * hw] - M) / se;
        }
        conf_out[(long long)b * hw + pix] = conf;
    }
}

}  // namespace mvd

extern "C" int mvd_softmax_regress_f32(const float* cost, const float* depth_values, int B, int D, int h, int w,
                                       float* depth_out, float* conf_out, mvd_stream_t stream) {
    MVD_REQUIRE(cost && depth_values && depth_out, "softmax_regress: NULL argument");
    MVD_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0 && B <= 65535, "softmax_regress: bad dimension");
    const long long hw = (long long)h * w;
    dim3 grid((unsigned)((hw + 63) / 64), (unsigned)B);
    hipLaunchKernelGGL(mvd::softmax_regress_kernel, grid, dim3(256), 0, (hipStream_t)stream, cost, depth_values, D, hw,
                       depth_out, conf_out);
    return mvd::launch_status("softmax_regress");
}
