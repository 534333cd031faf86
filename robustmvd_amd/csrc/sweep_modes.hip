// Other consumers of the plane sweep as reduction modes of one generic kernel:
//   MVD_REDUCE_VARIANCE        MVSNet's variance over key + V sources             rmvd/models/mvsnet.py:124-135
//   MVD_REDUCE_VARIANCE_KEYSQ  CVP-MVSNet's cost volume as the reference computes it: `volume_sum = ref_volume;
//                              volume_sq_sum = ref_volume.pow_(2)` alias one tensor, so the running SUM starts from key^2 too
//                              (rmvd/models/cvp_mvsnet.py:129-130, blocks/cvp_mvsnet_components.py:393-394)
//   MVD_REDUCE_GROUPCORR       Vis-MVSNet's group-wise correlation, one volume per source view
//                              (rmvd/models/blocks/utils.py:71-89 called from blocks/vis_mvsnet_singlestage.py:242)
// with the two things those consumers vary: PER-PIXEL depth hypotheses (B,D,h,w) (cvp proj_cost, cvp_mvsnet_components.py:
// 375-456; vis depth_start n1hw) and the pixel-centre convention of the warp (`pix_offset`, `scale`, `bias`:
// homo_warp / homo_warping sample at (X/Z) * W/(W-1) - 0.5 from integer pixel positions, blocks/utils.py:246-264;
// homography_warping samples at X/Z - 0.5 from positions x + 0.5, blocks/utils.py:154-186).
// Per view the caller passes the 3x4 matrix [R | t] with (X,Y,Z) = R (x+o, y+o, 1)^T d + t.
//
// Generic and simple by design (the tuned, marching K3 is the hot path): thread = (pixel, unit), unit = channel quad
// (variance modes) or channel group (group correlation); pixels are the fast index so that stores into the
// reference's (B,C,D,h,w) layout are coalesced.  Gathers come from zero-bordered channel-last copies like K3's.
#include "mvd_common.h"

namespace mvd {
int repack_padded_launch(const float* src, float* dst, int B, int C, int h, int w, hipStream_t st);
size_t padded_slot_bytes_public(int B, int C, int h, int w);

struct ReduceParams {
    ViewPtrs src;        // V x (B,h+3,w+3,C) zero-bordered channel-last
    ViewPtrs M;          // V x (B,3,4)
    ViewOutPtrs out;     // variance: out[0] (B,C,D,h,w); group correlation: out[v] (B,G,D,h,w)
    const float* key;    // (B,h+3,w+3,C)
    const float* depth;  // (B,D) or (B,D,h,w)
    int depth_per_pixel;
    float pix_offset, scale_x, scale_y, bias;
    int mode, groups;
    int B, C, D, h, w, V;
};

// One bilinear sample of 4 channels of source view v at the position plane `depth` puts key pixel (fx, fy) at.
__device__ __forceinline__ float4 reduce_sample(const ReduceParams& p, int v, int b, float fx, float fy, float depth, int c0, int W2,
                                                size_t img) {
    const float xhi = (float)p.w, yhi = (float)p.h;
    const int C = p.C;
    const float* __restrict__ M = p.M.p[v] + (size_t)b * 12;
    const float ax = fmaf(M[0], fx, fmaf(M[1], fy, M[2])), ay = fmaf(M[4], fx, fmaf(M[5], fy, M[6]));
    const float az = fmaf(M[8], fx, fmaf(M[9], fy, M[10]));
    const float X = fmaf(ax, depth, M[3]), Y = fmaf(ay, depth, M[7]), Z = fmaf(az, depth, M[11]);
    float ix = fmaf(X / Z, p.scale_x, p.bias), iy = fmaf(Y / Z, p.scale_y, p.bias);
    ix = __builtin_amdgcn_fmed3f(ix, -1.0f, xhi);  // NaN -> -1: all taps in the zero border
    iy = __builtin_amdgcn_fmed3f(iy, -1.0f, yhi);
    const float xf = floorf(ix), yf = floorf(iy);
    const float wx = ix - xf, wy = iy - yf, ux = 1.0f - wx, uy = 1.0f - wy;
    const float* __restrict__ f = p.src.p[v] + b * img + ((size_t)((int)yf + 1) * W2 + ((int)xf + 1)) * C + c0;
    const float4 a = *reinterpret_cast<const float4*>(f), bq = *reinterpret_cast<const float4*>(f + C);
    const float4 c = *reinterpret_cast<const float4*>(f + (size_t)W2 * C);
    const float4 e = *reinterpret_cast<const float4*>(f + (size_t)W2 * C + C);
    const float w00 = ux * uy, w10 = wx * uy, w01 = ux * wy, w11 = wx * wy;
    return make_float4(fmaf(e.x, w11, fmaf(c.x, w01, fmaf(bq.x, w10, a.x * w00))),
                       fmaf(e.y, w11, fmaf(c.y, w01, fmaf(bq.y, w10, a.y * w00))),
                       fmaf(e.z, w11, fmaf(c.z, w01, fmaf(bq.z, w10, a.z * w00))),
                       fmaf(e.w, w11, fmaf(c.w, w01, fmaf(bq.w, w10, a.w * w00))));
}

// The same reduction with the lanes of a pixel side by side (unit fastest: a pixel's units read whole 128-byte lines of a tap,
// the plain kernel's lanes = pixels read 16 bytes of each) and the results of a plane turned through LDS, so that a channel's
// 256 / units consecutive pixels leave as one run.  units = a power of two <= 64 (C / 4, or the number of groups).
// grid (ceil(h w / ppw), B), ppw = 256 / units pixels per workgroup.  Same arithmetic as sweep_reduce_kernel, bit for bit.
__global__ void __launch_bounds__(256) sweep_reduce_tile_kernel(ReduceParams p, int units) {
    __shared__ __attribute__((aligned(16))) float tile[2][1024];
    const int h = p.h, w = p.w, C = p.C, D = p.D, V = p.V;
    const bool corr = p.mode == MVD_REDUCE_GROUPCORR;
    const int qpu = corr ? C / p.groups / 4 : 1;
    const int ppw = 256 / units;
    const int tid = threadIdx.x, unit = tid % units, lp = tid / units;
    const int b = blockIdx.y;
    const long long npix = (long long)h * w, pix0 = (long long)blockIdx.x * ppw;
    const long long pix = min(pix0 + lp, npix - 1);  // lanes beyond the map repeat its last pixel; their results are not stored
    const int x = (int)(pix % w), y = (int)(pix / w);
    const int W2 = w + 3;
    const size_t img = (size_t)(h + 3) * W2 * C;
    const float fx = (float)x + p.pix_offset, fy = (float)y + p.pix_offset;
    const float inv_nv = 1.0f / (float)(V + 1);
    const size_t dplane = (size_t)h * w;
    const bool vec4 = (dplane % 4 == 0) && ppw >= 4 && ((size_t)p.out.p[0] & 15) == 0;
    int buf = 0;

    for (int d = 0; d < D; ++d) {
        const float depth = p.depth_per_pixel ? p.depth[(((size_t)b * D + d) * h + y) * w + x] : p.depth[(size_t)b * D + d];
        if (!corr) {
            const int c0 = unit * 4;
            const float4 k = *reinterpret_cast<const float4*>(p.key + b * img + ((size_t)(y + 1) * W2 + (x + 1)) * C + c0);
            float4 s1, s2;
            s2 = make_float4(k.x * k.x, k.y * k.y, k.z * k.z, k.w * k.w);
            s1 = p.mode == MVD_REDUCE_VARIANCE_KEYSQ ? s2 : k;  // the reference's aliasing: both sums start from key^2
            for (int v = 0; v < V; ++v) {
                const float4 sv = reduce_sample(p, v, b, fx, fy, depth, c0, W2, img);
                s1.x += sv.x; s1.y += sv.y; s1.z += sv.z; s1.w += sv.w;
                s2.x = fmaf(sv.x, sv.x, s2.x); s2.y = fmaf(sv.y, sv.y, s2.y);
                s2.z = fmaf(sv.z, sv.z, s2.z); s2.w = fmaf(sv.w, sv.w, s2.w);
            }
            const float mx = s1.x * inv_nv, my = s1.y * inv_nv, mz = s1.z * inv_nv, mw = s1.w * inv_nv;
            float* t = tile[buf] + (c0 * ppw + lp);  // [channel][pixel]
            t[0] = fmaf(s2.x, inv_nv, -mx * mx);
            t[ppw] = fmaf(s2.y, inv_nv, -my * my);
            t[2 * ppw] = fmaf(s2.z, inv_nv, -mz * mz);
            t[3 * ppw] = fmaf(s2.w, inv_nv, -mw * mw);
            __syncthreads();
            // 1024 results = C channels x ppw pixels: thread -> 4 consecutive pixels of one channel
            if (vec4) {
                const int l4 = ppw / 4, ch = tid / l4, p4 = (tid % l4) * 4;
                const long long po = pix0 + p4;
                float* o = p.out.p[0] + (((size_t)b * C + ch) * D + d) * dplane + po;
                const float4 r = *reinterpret_cast<const float4*>(tile[buf] + ch * ppw + p4);
                if (po + 3 < npix) *reinterpret_cast<float4*>(o) = r;
                else {
                    if (po < npix) o[0] = r.x;
                    if (po + 1 < npix) o[1] = r.y;
                    if (po + 2 < npix) o[2] = r.z;
                }
            } else {
                for (int e = tid; e < 1024; e += 256) {
                    const int ch = e / ppw, pp = e % ppw;
                    if (pix0 + pp < npix) p.out.p[0][(((size_t)b * C + ch) * D + d) * dplane + pix0 + pp] = tile[buf][e];
                }
            }
            buf ^= 1;
        } else {
            for (int v = 0; v < V; ++v) {
                float acc = 0.0f;
                for (int qq = 0; qq < qpu; ++qq) {
                    const int c0 = (unit * qpu + qq) * 4;
                    const float4 k = *reinterpret_cast<const float4*>(p.key + b * img + ((size_t)(y + 1) * W2 + (x + 1)) * C + c0);
                    const float4 sv = reduce_sample(p, v, b, fx, fy, depth, c0, W2, img);
                    const float dot = fmaf(k.w, sv.w, fmaf(k.z, sv.z, fmaf(k.y, sv.y, k.x * sv.x)));
                    acc = (qq == 0 ? 0.0f : acc) + dot;
                }
                tile[buf][unit * ppw + lp] = acc;  // [group][pixel]
                __syncthreads();
                const int g = tid / ppw, pp = tid % ppw;
                if (pix0 + pp < npix) p.out.p[v][(((size_t)b * p.groups + g) * D + d) * dplane + pix0 + pp] = tile[buf][tid];
                buf ^= 1;
            }
        }
    }
}

__global__ void __launch_bounds__(256) sweep_reduce_kernel(ReduceParams p) {
    const int h = p.h, w = p.w, C = p.C, D = p.D, V = p.V;
    const bool corr = p.mode == MVD_REDUCE_GROUPCORR;
    const int units = corr ? p.groups : C / 4;      // units per pixel
    const int qpu = corr ? C / p.groups / 4 : 1;    // channel quads per unit
    const long long npix = (long long)h * w;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)p.B * units * npix) return;
    const long long pix = t % npix;
    const int unit = (int)((t / npix) % units), b = (int)(t / (npix * units));
    const int x = (int)(pix % w), y = (int)(pix / w);
    const int W2 = w + 3;
    const size_t img = (size_t)(h + 3) * W2 * C;
    const float fx = (float)x + p.pix_offset, fy = (float)y + p.pix_offset;
    const float xhi = (float)w, yhi = (float)h;
    const float inv_nv = 1.0f / (float)(V + 1);
    const size_t dplane = (size_t)h * w;

    for (int d = 0; d < D; ++d) {
        const float depth = p.depth_per_pixel ? p.depth[(((size_t)b * D + d) * h + y) * w + x] : p.depth[(size_t)b * D + d];
        for (int qq = 0; qq < qpu; ++qq) {
            const int c0 = (unit * qpu + qq) * 4;
            const float4 k = *reinterpret_cast<const float4*>(p.key + b * img + ((size_t)(y + 1) * W2 + (x + 1)) * C + c0);
            float4 s1, s2;
            if (p.mode == MVD_REDUCE_VARIANCE_KEYSQ) {  // the reference's aliasing: sum and sum of squares both start from key^2
                s2 = make_float4(k.x * k.x, k.y * k.y, k.z * k.z, k.w * k.w);
                s1 = s2;
            } else {
                s1 = k;
                s2 = make_float4(k.x * k.x, k.y * k.y, k.z * k.z, k.w * k.w);
            }
            for (int v = 0; v < V; ++v) {
                const float* __restrict__ M = p.M.p[v] + (size_t)b * 12;
                const float ax = fmaf(M[0], fx, fmaf(M[1], fy, M[2])), ay = fmaf(M[4], fx, fmaf(M[5], fy, M[6]));
                const float az = fmaf(M[8], fx, fmaf(M[9], fy, M[10]));
                const float X = fmaf(ax, depth, M[3]), Y = fmaf(ay, depth, M[7]), Z = fmaf(az, depth, M[11]);
                float ix = fmaf(X / Z, p.scale_x, p.bias), iy = fmaf(Y / Z, p.scale_y, p.bias);
                ix = __builtin_amdgcn_fmed3f(ix, -1.0f, xhi);  // NaN -> -1: all taps in the zero border
                iy = __builtin_amdgcn_fmed3f(iy, -1.0f, yhi);
                const float xf = floorf(ix), yf = floorf(iy);
                const float wx = ix - xf, wy = iy - yf, ux = 1.0f - wx, uy = 1.0f - wy;
                const float* __restrict__ f = p.src.p[v] + b * img + ((size_t)((int)yf + 1) * W2 + ((int)xf + 1)) * C + c0;
                const float4 a = *reinterpret_cast<const float4*>(f), bq = *reinterpret_cast<const float4*>(f + C);
                const float4 c = *reinterpret_cast<const float4*>(f + (size_t)W2 * C);
                const float4 e = *reinterpret_cast<const float4*>(f + (size_t)W2 * C + C);
                const float w00 = ux * uy, w10 = wx * uy, w01 = ux * wy, w11 = wx * wy;
                const float4 sv = make_float4(fmaf(e.x, w11, fmaf(c.x, w01, fmaf(bq.x, w10, a.x * w00))),
                                              fmaf(e.y, w11, fmaf(c.y, w01, fmaf(bq.y, w10, a.y * w00))),
                                              fmaf(e.z, w11, fmaf(c.z, w01, fmaf(bq.z, w10, a.z * w00))),
                                              fmaf(e.w, w11, fmaf(c.w, w01, fmaf(bq.w, w10, a.w * w00))));
                if (corr) {
                    const float dot = fmaf(k.w, sv.w, fmaf(k.z, sv.z, fmaf(k.y, sv.y, k.x * sv.x)));
                    float* o = p.out.p[v] + (((size_t)b * p.groups + unit) * D + d) * dplane + pix;
                    *o = (qq == 0 ? 0.0f : *o) + dot;
                } else {
                    s1.x += sv.x; s1.y += sv.y; s1.z += sv.z; s1.w += sv.w;
                    s2.x = fmaf(sv.x, sv.x, s2.x); s2.y = fmaf(sv.y, sv.y, s2.y);
                    s2.z = fmaf(sv.z, sv.z, s2.z); s2.w = fmaf(sv.w, sv.w, s2.w);
                }
            }
            if (!corr) {
                const float mx = s1.x * inv_nv, my = s1.y * inv_nv, mz = s1.z * inv_nv, mw = s1.w * inv_nv;
                float* o = p.out.p[0] + (((size_t)b * C + c0) * D + d) * dplane + pix;
                o[0] = fmaf(s2.x, inv_nv, -mx * mx);
                o[(size_t)D * dplane] = fmaf(s2.y, inv_nv, -my * my);
                o[2 * (size_t)D * dplane] = fmaf(s2.z, inv_nv, -mz * mz);
                o[3 * (size_t)D * dplane] = fmaf(s2.w, inv_nv, -mw * mw);
            }
        }
    }
}

}  // namespace mvd

extern "C" {

size_t mvd_sweep_reduce_workspace_bytes(int B, int C, int h, int w, int V) {
    if (B <= 0 || C <= 0 || h <= 0 || w <= 0 || V < 0) return 0;
    return (size_t)(V + 1) * mvd::padded_slot_bytes_public(B, C, h, w);
}

int mvd_sweep_reduce_f32(const float* key_feat, const float* const* src_feat, const float* const* M, const float* depth,
                         int depth_per_pixel, float pix_offset, float scale_x, float scale_y, float bias, int mode, int groups,
                         int B, int C, int D, int h, int w, int V, float* const* out, void* workspace, size_t workspace_bytes,
                         mvd_stream_t stream) {
    using namespace mvd;
    MVD_REQUIRE(key_feat && src_feat && M && depth && out, "sweep_reduce: NULL argument");
    MVD_REQUIRE(B > 0 && D > 0 && h > 1 && w > 1 && V >= 1 && V <= MVD_MAX_VIEWS, "sweep_reduce: bad dimensions");
    MVD_REQUIRE(C >= 4 && C % 4 == 0, "sweep_reduce: C=%d must be a positive multiple of 4", C);
    MVD_REQUIRE(mode == MVD_REDUCE_VARIANCE || mode == MVD_REDUCE_VARIANCE_KEYSQ || mode == MVD_REDUCE_GROUPCORR, "sweep_reduce: mode %d", mode);
    if (mode == MVD_REDUCE_GROUPCORR)
        MVD_REQUIRE(groups > 0 && C % groups == 0 && (C / groups) % 4 == 0, "sweep_reduce: C/groups = %d/%d must be a multiple of 4", C, groups);
    const size_t need = mvd_sweep_reduce_workspace_bytes(B, C, h, w, V);
    if (!workspace || workspace_bytes < need) {
        set_error("sweep_reduce: workspace %zu B < required %zu B", workspace_bytes, need);
        return MVD_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t slot = padded_slot_bytes_public(B, C, h, w);
    char* ws = (char*)workspace;
    ReduceParams p{};
    int rc = repack_padded_launch(key_feat, (float*)ws, B, C, h, w, st);
    if (rc) return rc;
    p.key = (float*)ws;
    ws += slot;
    const int nout = mode == MVD_REDUCE_GROUPCORR ? V : 1;
    for (int v = 0; v < V; ++v) {
        MVD_REQUIRE(src_feat[v] && M[v], "sweep_reduce: NULL view %d", v);
        rc = repack_padded_launch(src_feat[v], (float*)ws, B, C, h, w, st);
        if (rc) return rc;
        p.src.p[v] = (float*)ws;
        ws += slot;
        p.M.p[v] = M[v];
    }
    for (int v = 0; v < nout; ++v) {
        MVD_REQUIRE(out[v], "sweep_reduce: NULL output %d", v);
        p.out.p[v] = out[v];
    }
    p.depth = depth; p.depth_per_pixel = depth_per_pixel;
    p.pix_offset = pix_offset; p.scale_x = scale_x; p.scale_y = scale_y; p.bias = bias;
    p.mode = mode; p.groups = groups;
    p.B = B; p.C = C; p.D = D; p.h = h; p.w = w; p.V = V;
    const int units = mode == MVD_REDUCE_GROUPCORR ? groups : C / 4;
    const long long nthr = (long long)B * units * h * w, nblk = (nthr + 255) / 256;
    MVD_REQUIRE(nblk <= 0x7fffffffLL, "sweep_reduce: grid too large");
    // (experiments library, MVD_REDUCE_PLAIN: the pixel-per-lane kernel for every shape — the variants test compares the two)
    if (units >= 2 && units <= 64 && (units & (units - 1)) == 0 && B <= 65535 && !exp_env("MVD_REDUCE_PLAIN")) {
        const int ppw = 256 / units;
        const long long nbx = ((long long)h * w + ppw - 1) / ppw;
        MVD_REQUIRE(nbx <= 0x7fffffffLL, "sweep_reduce: grid too large");
        hipLaunchKernelGGL(sweep_reduce_tile_kernel, dim3((unsigned)nbx, (unsigned)B), dim3(256), 0, st, p, units);
        return launch_status("sweep_reduce_tile");
    }
    hipLaunchKernelGGL(sweep_reduce_kernel, dim3((unsigned)nblk), dim3(256), 0, st, p);
    return launch_status("sweep_reduce");
}
}
