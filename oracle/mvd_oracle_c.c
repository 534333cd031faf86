/*
 * CPU ORACLE, C/OpenMP part — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's plane-sweep hot path (same formulas as oracle/mvd_oracle.py,
 * which is the readable form; this file exists so that the oracle finishes mid-size cases in seconds
 * and so that bench.py's cpu_baseline leg times a multi-threaded CPU port on the GPU box's host cores).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the resulting
 * oracle/_build/libmvd_oracle.so; the product never does.
 *
 * Parity pinning: tests/test_oracle_golden.py checks every function below against the golden vectors
 * generated from the reference's own CPU path (tests/golden/g*.npz).
 * Citations are file:line under the reference repository root.
 * Layouts are the reference's: NCHW features, (B,C,D,h,w) volumes, all float32.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* ---- shared: bilinear taps with zero padding (ATen grid_sampler_2d, align_corners=False) ---- */
typedef struct {
    int off[4];
    float w[4];
    float inb;
} taps_t;

static inline float unnorm(float g, float size) { return ((g + 1.0f) * size - 1.0f) / 2.0f; }

static inline taps_t bilinear_taps(float ix, float iy, int hs, int ws) {
    taps_t t;
    const float x0 = floorf(ix), y0 = floorf(iy), x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    const float xs[4] = {x0, x1, x0, x1}, ys[4] = {y0, y0, y1, y1};
    const float wt[4] = {(x1 - ix) * (y1 - iy), (ix - x0) * (y1 - iy), (x1 - ix) * (iy - y0), (ix - x0) * (iy - y0)};
    t.inb = 0.0f;
    for (int k = 0; k < 4; ++k) {
        const int in = xs[k] >= 0.0f && xs[k] <= (float)(ws - 1) && ys[k] >= 0.0f && ys[k] <= (float)(hs - 1);
        t.off[k] = in ? (int)ys[k] * ws + (int)xs[k] : 0;
        t.w[k] = in ? wt[k] : 0.0f;
        t.inb += t.w[k];
    }
    return t;
}

/* ============================ Path B ======================================================== */

/* homo_warp (blocks/utils.py:222-268) for V views + variance (mvsnet.py:124-136).
 * key (B,C,h,w); srcs[v] (B,C,h,w); projs[v] (B,4,4); key_inv (B,4,4); depth (B,D); out (B,C,D,h,w).
 * If warped_only != 0: V must be 1 and out receives the warped volume instead. */
void orc_warp_variance(const float* key, const float* const* srcs, const float* const* projs, const float* key_inv,
                       const float* depth, int B, int C, int D, int h, int w, int V, int warped_only, float* out) {
    const size_t hw = (size_t)h * w;
    for (int b = 0; b < B; ++b) {
        float M[64][12]; /* V <= 64 */
        for (int v = 0; v < V; ++v) {
            const float* P = projs[v] + b * 16;
            const float* Q = key_inv + b * 16;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 4; ++j) {
                    float a = 0.0f;
                    for (int k = 0; k < 4; ++k) a += P[i * 4 + k] * Q[k * 4 + j];
                    M[v][i * 4 + j] = a;
                }
        }
#pragma omp parallel for collapse(2) schedule(static)
        for (int d = 0; d < D; ++d)
            for (int y = 0; y < h; ++y) {
                const float dep = depth[b * D + d];
                const float nv = (float)(V + 1);
                taps_t* tp = (taps_t*)malloc(sizeof(taps_t) * (size_t)w * V);
                for (int v = 0; v < V; ++v)
                    for (int x = 0; x < w; ++x) {
                        const float gx = (float)x * dep, gy = (float)y * dep;
                        const float* m = M[v];
                        const float X = m[0] * gx + m[1] * gy + m[2] * dep + m[3];
                        const float Y = m[4] * gx + m[5] * gy + m[6] * dep + m[7];
                        const float Z = m[8] * gx + m[9] * gy + m[10] * dep + m[11];
                        const float nx = (X / Z) / ((float)(w - 1) / 2.0f) - 1.0f; /* utils.py:256-257 */
                        const float ny = (Y / Z) / ((float)(h - 1) / 2.0f) - 1.0f;
                        tp[v * w + x] = bilinear_taps(unnorm(nx, (float)w), unnorm(ny, (float)h), h, w);
                    }
                for (int c = 0; c < C; ++c) {
                    float* o = out + (((size_t)b * C + c) * D + d) * hw + (size_t)y * w;
                    const float* kf = key + ((size_t)b * C + c) * hw + (size_t)y * w;
                    for (int x = 0; x < w; ++x) {
                        float s1 = warped_only ? 0.0f : kf[x];
                        float s2 = s1 * s1;
                        for (int v = 0; v < V; ++v) {
                            const float* sf = srcs[v] + ((size_t)b * C + c) * hw;
                            const taps_t* t = &tp[v * w + x];
                            float a = 0.0f;
                            for (int k = 0; k < 4; ++k) a += sf[t->off[k]] * t->w[k];
                            s1 += a;
                            s2 += a * a;
                        }
                        if (warped_only) o[x] = s1;
                        else {
                            const float mean = s1 / nv;
                            o[x] = s2 / nv - mean * mean;
                        }
                    }
                }
                free(tp);
            }
    }
}

/* 3x3x3 Conv3d, padding 1, stride 1 or 2 (ConvBnReLU3D, mvsnet_components.py:25-41), fused with the
 * eval-mode BatchNorm as y = relu?(conv*scale + shift) (+ skip).  x (Cin,D,h,w), wgt (Cout,Cin,3,3,3). */
void orc_conv3d(const float* x, const float* wgt, const float* scale, const float* shift, const float* skip,
                int Cin, int Cout, int D, int h, int w, int stride, int relu, float* y) {
    const int Do = stride == 1 ? D : D / 2, ho = stride == 1 ? h : h / 2, wo = stride == 1 ? w : w / 2;
    const size_t ihw = (size_t)h * w, ohw = (size_t)ho * wo;
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Cout; ++co)
        for (int od = 0; od < Do; ++od) {
            float* acc = (float*)calloc(ohw, sizeof(float));
            for (int ci = 0; ci < Cin; ++ci)
                for (int kd = 0; kd < 3; ++kd) {
                    const int id = od * stride + kd - 1;
                    if (id < 0 || id >= D) continue;
                    const float* xp = x + ((size_t)ci * D + id) * ihw;
                    for (int kh = 0; kh < 3; ++kh)
                        for (int kw = 0; kw < 3; ++kw) {
                            const float wv = wgt[(((size_t)co * Cin + ci) * 27) + (kd * 3 + kh) * 3 + kw];
                            for (int oh = 0; oh < ho; ++oh) {
                                const int ih = oh * stride + kh - 1;
                                if (ih < 0 || ih >= h) continue;
                                const float* xr = xp + (size_t)ih * w;
                                float* ar = acc + (size_t)oh * wo;
                                int lo = 0, hi = wo;
                                while (lo < wo && lo * stride + kw - 1 < 0) ++lo;
                                while (hi > lo && (hi - 1) * stride + kw - 1 >= w) --hi;
                                if (stride == 1)
                                    for (int ow = lo; ow < hi; ++ow) ar[ow] += wv * xr[ow + kw - 1];
                                else
                                    for (int ow = lo; ow < hi; ++ow) ar[ow] += wv * xr[2 * ow + kw - 1];
                            }
                        }
                }
            float* yo = y + ((size_t)co * Do + od) * ohw;
            const float* sk = skip ? skip + ((size_t)co * Do + od) * ohw : NULL;
            for (size_t i = 0; i < ohw; ++i) {
                float v = acc[i] * scale[co] + shift[co];
                if (relu && v < 0.0f) v = 0.0f;
                if (sk) v += sk[i];
                yo[i] = v;
            }
            free(acc);
        }
}

/* ConvTranspose3d k3 s2 p1 output_padding 1 (mvsnet_components.py:84-109): out[o] += x[i]*w[k], o = 2i-1+k.
 * x (Cin,D,h,w), wgt (Cin,Cout,3,3,3) -> y (Cout,2D,2h,2w); same fused epilogue as orc_conv3d. */
void orc_deconv3d(const float* x, const float* wgt, const float* scale, const float* shift, const float* skip,
                  int Cin, int Cout, int D, int h, int w, int relu, float* y) {
    const int Do = 2 * D, ho = 2 * h, wo = 2 * w;
    const size_t ihw = (size_t)h * w, ohw = (size_t)ho * wo;
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Cout; ++co)
        for (int od = 0; od < Do; ++od) {
            float* acc = (float*)calloc(ohw, sizeof(float));
            for (int kd = 0; kd < 3; ++kd) {
                const int t = od + 1 - kd;
                if (t < 0 || (t & 1) || t / 2 >= D) continue;
                const int id = t / 2;
                for (int ci = 0; ci < Cin; ++ci) {
                    const float* xp = x + ((size_t)ci * D + id) * ihw;
                    for (int kh = 0; kh < 3; ++kh)
                        for (int kw = 0; kw < 3; ++kw) {
                            const float wv = wgt[(((size_t)ci * Cout + co) * 27) + (kd * 3 + kh) * 3 + kw];
                            for (int ih = 0; ih < h; ++ih) {
                                const int oh = 2 * ih - 1 + kh;
                                if (oh < 0 || oh >= ho) continue;
                                const float* xr = xp + (size_t)ih * w;
                                float* ar = acc + (size_t)oh * wo;
                                for (int iw = 0; iw < w; ++iw) {
                                    const int ow = 2 * iw - 1 + kw;
                                    if (ow >= 0 && ow < wo) ar[ow] += wv * xr[iw];
                                }
                            }
                        }
                }
            }
            float* yo = y + ((size_t)co * Do + od) * ohw;
            const float* sk = skip ? skip + ((size_t)co * Do + od) * ohw : NULL;
            for (size_t i = 0; i < ohw; ++i) {
                float v = acc[i] * scale[co] + shift[co];
                if (relu && v < 0.0f) v = 0.0f;
                if (sk) v += sk[i];
                yo[i] = v;
            }
            free(acc);
        }
}

/* softmax + depth regression + 4-bin confidence (mvsnet.py:139-160). cost (B,D,h,w), depth (B,D). */
void orc_softmax_regress(const float* cost, const float* depth, int B, int D, int h, int w, float* depth_out,
                         float* conf_out) {
    const size_t hw = (size_t)h * w;
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)(B * hw); ++i) {
        const int b = (int)(i / hw);
        const size_t p = (size_t)(i % hw);
        const float* c = cost + (size_t)b * D * hw + p;
        float m = -INFINITY;
        for (int d = 0; d < D; ++d) m = fmaxf(m, c[d * hw]);
        float se = 0.0f;
        for (int d = 0; d < D; ++d) se += expf(c[d * hw] - m);
        float dep = 0.0f, fi = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float pd = expf(c[d * hw] - m) / se;
            dep += pd * depth[b * D + d];
            fi += pd * (float)d;
        }
        depth_out[i] = dep;
        const int idx = (int)fi;
        float cf = 0.0f;
        for (int j = idx - 1; j <= idx + 2; ++j)
            if (j >= 0 && j < D) cf += expf(c[j * hw] - m) / se;
        conf_out[i] = cf;
    }
}

/* ============================ Path A ======================================================== */

/* PlanesweepCorrelation for one source view (planesweep_corr.py:228-349,489-521,152-195).
 * fk (N,C,h,w), fs (N,C,hs,ws), Kk/Ks (N,3,3) relative intrinsics, T (N,4,4), invd (N or 1, S). */
void orc_sweep_corr(const float* fk, const float* fs, const float* Kk, const float* Ks, const float* T,
                    const float* invd, int invd_batched, int N, int C, int h, int w, int hs, int ws, int S,
                    float* corr, float* mask) {
    const size_t khw = (size_t)h * w, shw = (size_t)hs * ws;
    const float isc = 1.0f / sqrtf((float)C);
    for (int n = 0; n < N; ++n) {
        const float* K = Kk + n * 9;
        const float* Ko = Ks + n * 9;
        const float* t = T + n * 16;
        const float fx = K[0] * (float)w, fy = K[4] * (float)h, cx = K[2] * (float)w, cy = K[5] * (float)h;
        const float fxo = Ko[0] * (float)ws, fyo = Ko[4] * (float)hs, cxo = Ko[2] * (float)ws, cyo = Ko[5] * (float)hs;
        const float r11 = t[0], r12 = t[1], r13 = t[2], t1 = t[3], r21 = t[4], r22 = t[5], r23 = t[6], t2 = t[7];
        const float r31 = t[8], r32 = t[9], r33 = t[10], t3 = t[11];
        const float A = fxo * r11 + cxo * r31, Bq = fxo * r12 + cxo * r32;
        const float a = A / fx, b = Bq / fy, c = -(cx * A / fx) - (cy * Bq / fy) + (fxo * r13 + cxo * r33);
        const float e = fxo * t1 + cxo * t3;
        const float F = fyo * r21 + cyo * r31, G = fyo * r22 + cyo * r32;
        const float f = F / fx, g = G / fy, hh = -(cx * F / fx) - (cy * G / fy) + (fyo * r23 + cyo * r33);
        const float ii = fyo * t2 + cyo * t3;
        const float j = r31 / fx, k = r32 / fy, l = -cx * r31 / fx - cy * r32 / fy + r33, m = t3;
        const float* inv = invd + (invd_batched ? (size_t)n * S : 0);
        const float* fkn = fk + (size_t)n * C * khw;
        const float* fsn = fs + (size_t)n * C * shw;
#pragma omp parallel for collapse(2) schedule(static)
        for (int s = 0; s < S; ++s)
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    const float xc = (float)x + 0.5f, yc = (float)y + 0.5f;
                    const float u_inf = (a * xc + b * yc) + c, v_inf = (f * xc + g * yc) + hh, k_inf = (j * xc + k * yc) + l;
                    const float ds = inv[s], den = k_inf + m * ds;
                    float us = (u_inf + e * ds) / den, vs = (v_inf + ii * ds) / den;
                    if (isinf(us)) us = us > 0 ? 1e9f : -1e9f;
                    if (isnan(us)) us = 1e9f;
                    if (isinf(vs)) vs = vs > 0 ? 1e9f : -1e9f;
                    if (isnan(vs)) vs = 1e9f;
                    const float zs = 1.0f / ds, zp = -(m / k_inf);
                    const int vis = (zs > 0.0f) && ((k_inf > 0.0f && zs > zp) || (k_inf < 0.0f && zs < zp) || (k_inf == 0.0f && m > 0.0f));
                    const taps_t tp = bilinear_taps(unnorm(2.0f * us / (float)ws - 1.0f, (float)ws),
                                                    unnorm(2.0f * vs / (float)hs - 1.0f, (float)hs), hs, ws);
                    float acc = 0.0f;
                    for (int q = 0; q < 4; ++q) {
                        if (tp.w[q] == 0.0f) continue;
                        float dot = 0.0f;
                        const float* sp = fsn + tp.off[q];
                        const float* kp = fkn + (size_t)y * w + x;
                        for (int ch = 0; ch < C; ++ch) dot += kp[ch * khw] * sp[ch * shw];
                        acc += dot * isc * tp.w[q];
                    }
                    const float mk = (tp.inb < 0.9999f || !vis) ? 0.0f : 1.0f;
                    const size_t o = (((size_t)n * S + s) * h + y) * w + x;
                    corr[o] = acc * mk;
                    mask[o] = mk;
                }
    }
}

/* LearnedFusion arithmetic (learned_fusion.py:32-48) given the score maps. */
void orc_fuse_views(const float* const* corr, const float* const* mask, const float* const* score, int N, int S,
                    int h, int w, int V, float* fused, float* fmask) {
    const size_t hw = (size_t)h * w;
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)(N * hw); ++i) {
        const int n = (int)(i / hw);
        const size_t p = (size_t)(i % hw);
        float wv[64];
        float mx = -INFINITY, se = 0.0f;
        for (int v = 0; v < V; ++v) mx = fmaxf(mx, score[v][i]);
        for (int v = 0; v < V; ++v) { wv[v] = expf(score[v][i] - mx); se += wv[v]; }
        for (int v = 0; v < V; ++v) wv[v] = wv[v] / se + 1e-9f;
        for (int s = 0; s < S; ++s) {
            const size_t o = ((size_t)n * S + s) * hw + p;
            float ws_ = 0.0f, cs = 0.0f;
            for (int v = 0; v < V; ++v) {
                const float vw = wv[v] * mask[v][o];
                ws_ += vw;
                cs += corr[v][o] * vw;
            }
            const float fm = ws_ != 0.0f ? 1.0f : 0.0f;
            fused[o] = cs / (ws_ + 1e-9f) * fm;
            fmask[o] = fm;
        }
    }
}

int orc_num_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
