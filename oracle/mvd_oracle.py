"""CPU ORACLE (test infrastructure, NOT product code) — numpy restatement of the reference's
plane-sweep cost-volume hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product (robustmvd_amd/) never does and fails loudly when its HIP library is missing.

Parity pinning: the reference has no tests for this path, so every function here is checked in
tests/test_oracle_golden.py against golden vectors produced by running the reference's own CPU
path in the build container (tests/golden/make_golden.py, fixtures tests/golden/g*.npz).

All arithmetic is float32 and follows the reference's operation order where it matters for
bit-closeness.  Citations are file:line under /root/reference/.
"""
import numpy as np

F32 = np.float32


# =============================================================================================
# shared: bilinear sampling with zero padding, exactly as ATen grid_sampler_2d
# (mode=bilinear, padding_mode=zeros, align_corners=False) does it on a *normalised* grid.
# =============================================================================================
def unnormalize(g, size):
    """grid_sampler_unnormalize, align_corners=False: ((g + 1) * size - 1) / 2."""
    return ((g + F32(1.0)) * F32(size) - F32(1.0)) / F32(2.0)


def bilinear_taps(ix, iy, hs, ws):
    """Returns the 4 taps (x, y, weight, inbounds) in the order nw, ne, sw, se."""
    with np.errstate(invalid="ignore"):
        x0 = np.floor(ix)
        y0 = np.floor(iy)
        x1 = x0 + F32(1.0)
        y1 = y0 + F32(1.0)
        taps = []
        for xx, yy, wgt in (
            (x0, y0, (x1 - ix) * (y1 - iy)),
            (x1, y0, (ix - x0) * (y1 - iy)),
            (x0, y1, (x1 - ix) * (iy - y0)),
            (x1, y1, (ix - x0) * (iy - y0)),
        ):
            inb = (xx >= 0) & (xx <= ws - 1) & (yy >= 0) & (yy <= hs - 1)
            xi = np.where(inb, xx, 0).astype(np.int64)
            yi = np.where(inb, yy, 0).astype(np.int64)
            taps.append((xi, yi, wgt.astype(F32), inb))
    return taps


def grid_sample_zeros(img, ix, iy):
    """img (C,hs,ws); ix, iy arbitrary equal shapes (un-normalised source indices).
    Returns (C, *ix.shape).  NaN coordinates give NaN outputs only through in-bounds taps,
    i.e. never (a NaN comparison is false), matching ATen's CPU kernel."""
    C, hs, ws = img.shape
    out = np.zeros((C,) + ix.shape, F32)
    for xi, yi, wgt, inb in bilinear_taps(ix, iy, hs, ws):
        w_eff = np.where(inb, wgt, F32(0.0))
        out += img[:, yi, xi] * w_eff[None]
    return out


# =============================================================================================
# Path A — rows A1..A6
# =============================================================================================
def compute_sampling_invdepths(min_depth, max_depth, num_samples, sampling_type="linear_invdepth"):
    """A1. planesweep_corr.py:524-555.  Returns (N, S) float32, far -> near."""
    min_depth = np.atleast_1d(np.asarray(min_depth, F32))[:, None]
    max_depth = np.atleast_1d(np.asarray(max_depth, F32))[:, None]
    min_inv = F32(1.0) / max_depth
    max_inv = F32(1.0) / min_depth
    steps = np.arange(num_samples, dtype=F32)[None]
    if sampling_type == "linear_invdepth":
        return (min_inv + steps * (max_inv - min_inv) / F32(num_samples - 1)).astype(F32)
    if sampling_type == "linear_depth":
        inv = F32(1.0) / (min_depth + steps * (max_depth - min_depth) / F32(num_samples - 1))
        return inv[:, ::-1].astype(F32).copy()
    raise ValueError(sampling_type)


def epipolar_coeffs(K_key, K_src, T, h, w, hs, ws):
    """A2. EpipolarCoeffs.from_calib, planesweep_corr.py:228-300.
    K_*: (N,3,3) relative intrinsics; T: (N,4,4) source_to_key_transform (p_src = T p_key).
    Returns u_inf, v_inf, k_inf (N,h,w) and m_u, m_v, m_k (N,)."""
    K_key, K_src, T = (np.asarray(a, F32) for a in (K_key, K_src, T))
    x = (np.arange(w, dtype=F32) + F32(0.5))[None, None, :]
    y = (np.arange(h, dtype=F32) + F32(0.5))[None, :, None]
    c3 = lambda v: v[:, None, None]
    fx, fy = c3(K_key[:, 0, 0] * F32(w)), c3(K_key[:, 1, 1] * F32(h))
    cx, cy = c3(K_key[:, 0, 2] * F32(w)), c3(K_key[:, 1, 2] * F32(h))
    fxo, fyo = c3(K_src[:, 0, 0] * F32(ws)), c3(K_src[:, 1, 1] * F32(hs))
    cxo, cyo = c3(K_src[:, 0, 2] * F32(ws)), c3(K_src[:, 1, 2] * F32(hs))
    r = lambda i, j: c3(T[:, i, j])
    r11, r12, r13, t1 = r(0, 0), r(0, 1), r(0, 2), r(0, 3)
    r21, r22, r23, t2 = r(1, 0), r(1, 1), r(1, 2), r(1, 3)
    r31, r32, r33, t3 = r(2, 0), r(2, 1), r(2, 2), r(2, 3)
    a = (fxo * r11 + cxo * r31) / fx
    b = (fxo * r12 + cxo * r32) / fy
    c = -(cx * (fxo * r11 + cxo * r31) / fx) - (cy * (fxo * r12 + cxo * r32) / fy) + (fxo * r13 + cxo * r33)
    e = fxo * t1 + cxo * t3
    f = (fyo * r21 + cyo * r31) / fx
    g = (fyo * r22 + cyo * r32) / fy
    hh = -(cx * (fyo * r21 + cyo * r31) / fx) - (cy * (fyo * r22 + cyo * r32) / fy) + (fyo * r23 + cyo * r33)
    i = fyo * t2 + cyo * t3
    j = r31 / fx
    k = r32 / fy
    l = -cx * r31 / fx - cy * r32 / fy + r33
    m = t3
    u_inf = a * x + b * y + c
    v_inf = f * x + g * y + hh
    k_inf = j * x + k * y + l
    scal = dict(a=a, b=b, c=c, e=e, f=f, g=g, h=hh, i=i, j=j, k=k, l=l, m=m)
    return dict(u_inf=u_inf.astype(F32), v_inf=v_inf.astype(F32), k_inf=k_inf.astype(F32),
                m_u=e[:, 0, 0], m_v=i[:, 0, 0], m_k=m[:, 0, 0],
                scalars={n: v[:, 0, 0].astype(F32) for n, v in scal.items()})


def sweep_grids(co, invdepths):
    """A2+A3. us_from_ds / vs_from_ds with replace_nonfinite (planesweep_corr.py:333-349) and the
    visibility mask of get_plane_sweep_sampling_points (:489-512).
    invdepths (N,S) -> us, vs (N,S,h,w) float32; visible (N,S,h,w) bool."""
    ds = np.asarray(invdepths, F32)
    ds = ds[:, :, None, None] if ds.ndim == 2 else ds  # (N,S) or per key pixel (N,S,h,w) (planesweep_corr.py:465-487)
    u_inf, v_inf, k_inf = (co[n][:, None] for n in ("u_inf", "v_inf", "k_inf"))
    m_u, m_v, m_k = (co[n][:, None, None, None] for n in ("m_u", "m_v", "m_k"))
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        den = k_inf + m_k * ds
        us = (u_inf + m_u * ds) / den
        vs = (v_inf + m_v * ds) / den

        def fix(a):
            a = a.copy()
            inf = np.isinf(a)
            a[inf] = (F32(1e9) * np.sign(a))[inf]
            a[np.isnan(a)] = F32(1e9)
            return a

        us, vs = fix(us), fix(vs)
        zs = (F32(1.0) / ds) * np.ones_like(us)
        z_poles = -(m_k / k_inf)
        visible = (zs > 0) & (((k_inf > 0) & (zs > z_poles)) | ((k_inf < 0) & (zs < z_poles))
                               | ((k_inf == 0) & (m_k > 0)))
    return us.astype(F32), vs.astype(F32), visible


def sweep_corr_view(feat_key, feat_src, us, vs, visible, normalize="dim"):
    """A4. TorchCorr.forward (normalize="dim", zeros padding), planesweep_corr.py:152-195 with
    warp() :82-104.  The reference forms the all-pairs matrix and interpolates it; this is the
    equivalent warp-then-dot form (SURVEY.md 8c 'verified equivalence').
    feat_key (N,C,h,w), feat_src (N,C,hs,ws), us/vs/visible (N,S,h,w) -> corr, mask (N,S,h,w)."""
    N, C, h, w = feat_key.shape
    hs, ws = feat_src.shape[-2:]
    S = us.shape[1]
    corr = np.zeros((N, S, h, w), F32)
    mask = np.zeros((N, S, h, w), F32)
    inv_sqrt_c = F32(1.0) / np.sqrt(F32(C)) if normalize == "dim" else F32(1.0)
    if normalize and normalize != "dim":  # True / "before": x / (|x| + 1e-9) along C (planesweep_corr.py:8-10,165-167)
        feat_key = (feat_key / (np.linalg.norm(feat_key, axis=1, keepdims=True) + F32(1e-9))).astype(F32)
        feat_src = (feat_src / (np.linalg.norm(feat_src, axis=1, keepdims=True) + F32(1e-9))).astype(F32)
    for n in range(N):
        for s in range(S):
            gx = F32(2.0) * us[n, s] / F32(ws) - F32(1.0)  # warp(): :87-88
            gy = F32(2.0) * vs[n, s] / F32(hs) - F32(1.0)
            ix, iy = unnormalize(gx, ws), unnormalize(gy, hs)
            acc = np.zeros((h, w), F32)
            inb_sum = np.zeros((h, w), F32)
            for xi, yi, wgt, inb in bilinear_taps(ix, iy, hs, ws):
                w_eff = np.where(inb, wgt, F32(0.0))
                dots = np.einsum("chw,chw->hw", feat_key[n], feat_src[n][:, yi, xi]).astype(F32)
                acc += dots * inv_sqrt_c * w_eff
                inb_sum += w_eff
            m = np.where(inb_sum < F32(0.9999), F32(0.0), inb_sum)  # :101-102
            m = np.where(m > 0, F32(1.0), m)
            m = m * visible[n, s].astype(F32)
            corr[n, s] = acc * m
            mask[n, s] = m
    return corr, mask


def planesweep_correlation(feat_key, intrinsics_key, feat_sources, source_to_key_transforms,
                           intrinsics_sources=None, num_sampling_points=None, min_depth=None, max_depth=None,
                           sampling_invdepths=None, sampling_type="linear_invdepth", normalize="dim"):
    """A5. PlanesweepCorrelation.forward, planesweep_corr.py:396-427.  Same arguments, numpy in/out.
    Returns (corrs[V], masks[V], sampling_invdepths (N,S,1,1))."""
    N, C, h, w = feat_key.shape
    if intrinsics_sources is None:
        intrinsics_sources = [intrinsics_key] * len(feat_sources)
    assert len(feat_sources) == len(source_to_key_transforms) == len(intrinsics_sources)
    if min_depth is not None and max_depth is not None:
        assert sampling_invdepths is None and num_sampling_points is not None
        inv = compute_sampling_invdepths(min_depth, max_depth, num_sampling_points, sampling_type)
    else:
        assert num_sampling_points is None and sampling_invdepths is not None
        inv = np.asarray(sampling_invdepths, F32)
    per_pixel = inv.ndim == 4 and (inv.shape[2] > 1 or inv.shape[3] > 1)
    if per_pixel:
        inv_n = np.broadcast_to(inv, (N, inv.shape[1], h, w))
        inv_ret = inv
    else:
        inv = inv.reshape(inv.shape[0], inv.shape[1])
        inv_n = np.broadcast_to(inv, (N, inv.shape[1]))
        inv_ret = inv[:, :, None, None]
    corrs, masks = [], []
    for fs, T, Ks in zip(feat_sources, source_to_key_transforms, intrinsics_sources):
        co = epipolar_coeffs(intrinsics_key, Ks, T, h, w, fs.shape[2], fs.shape[3])
        us, vs, vis = sweep_grids(co, inv_n)
        c, m = sweep_corr_view(feat_key, fs, us, vs, vis, normalize)
        corrs.append(c)
        masks.append(m)
    return corrs, masks, inv_ret


def sweep_warp_view(feat_src, us, vs, normalize=False):
    """WarpOnlyCorr.forward (planesweep_corr.py:107-140) with warp_multi / warp (:13-104), zeros padding.
    feat_src (N,C,hs,ws), us/vs (N,S,h,w) -> warped (N,S,C,h,w), mask (N,S,h,w).  The mask is the SAMPLING mask only: the
    visibility mask that correlate() hands over (:515-517) is not used by this block."""
    N, C, hs, ws = feat_src.shape
    S, h, w = us.shape[1:]
    if normalize == "before":
        feat_src = (feat_src / (np.linalg.norm(feat_src, axis=1, keepdims=True) + F32(1e-9))).astype(F32)
    out = np.zeros((N, S, C, h, w), F32)
    mask = np.zeros((N, S, h, w), F32)
    for n in range(N):
        for s in range(S):
            gx = F32(2.0) * us[n, s] / F32(ws) - F32(1.0)
            gy = F32(2.0) * vs[n, s] / F32(hs) - F32(1.0)
            ix, iy = unnormalize(gx, ws), unnormalize(gy, hs)
            val = grid_sample_zeros(feat_src[n], ix, iy)  # (C,h,w)
            inb_sum = np.zeros((h, w), F32)
            for xi, yi, wgt, inb in bilinear_taps(ix, iy, hs, ws):
                inb_sum += np.where(inb, wgt, F32(0.0))
            m = np.where(inb_sum < F32(0.9999), F32(0.0), F32(1.0))
            if normalize and normalize not in ("before", "dim"):  # True / "after": along C (:135-136)
                val = (val / (np.linalg.norm(val, axis=0, keepdims=True) + F32(1e-9))).astype(F32)
            out[n, s] = val * m[None]
            mask[n, s] = m
    return out, mask


def planesweep_warp(feat_key_size, intrinsics_key, feat_sources, source_to_key_transforms, sampling_invdepths,
                    intrinsics_sources=None, normalize=False):
    """PlanesweepCorrelation(warp_only=True).forward, planesweep_corr.py:396-427 + 514-521: (warped[V], masks[V])."""
    h, w = feat_key_size
    N = feat_sources[0].shape[0]
    if intrinsics_sources is None:
        intrinsics_sources = [intrinsics_key] * len(feat_sources)
    inv = np.asarray(sampling_invdepths, F32)
    per_pixel = inv.ndim == 4 and (inv.shape[2] > 1 or inv.shape[3] > 1)
    inv_n = np.broadcast_to(inv, (N, inv.shape[1], h, w)) if per_pixel else np.broadcast_to(inv.reshape(inv.shape[0], inv.shape[1]), (N, inv.shape[1]))
    outs, masks = [], []
    for fs, T, Ks in zip(feat_sources, source_to_key_transforms, intrinsics_sources):
        co = epipolar_coeffs(intrinsics_key, Ks, T, h, w, fs.shape[2], fs.shape[3])
        us, vs, _vis = sweep_grids(co, inv_n)
        o, m = sweep_warp_view(fs, us, vs, normalize)
        outs.append(o)
        masks.append(m)
    return outs, masks


def conv2d(x, weight, bias=None, stride=1, padding=0):
    """Plain NCHW 2-D cross-correlation (used for the fusion score convs only)."""
    N, C, H, W = x.shape
    O, _, kh, kw = weight.shape
    xp = np.pad(x, ((0, 0), (0, 0), (padding, padding), (padding, padding)))
    Ho = (H + 2 * padding - kh) // stride + 1
    Wo = (W + 2 * padding - kw) // stride + 1
    cols = np.empty((N, C, kh, kw, Ho, Wo), F32)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, i, j] = xp[:, :, i:i + stride * Ho:stride, j:j + stride * Wo:stride]
    out = np.einsum("ncijhw,ocij->nohw", cols, weight, optimize=True).astype(F32)
    if bias is not None:
        out += bias[None, :, None, None]
    return out


def fusion_scores(corr, w0, b0, w1, b1):
    """corr_to_view_weight: conv3x3(256->128) + ReLU + conv1x1(128->1), learned_fusion.py:13-17."""
    hid = np.maximum(conv2d(corr, w0, b0, padding=1), 0)
    return conv2d(hid, w1, b1)


def fuse_views(corrs, masks, scores):
    """A6. LearnedFusion.forward, learned_fusion.py:24-54, given the per-view score maps
    scores[v] (N,1,h,w).  V == 1 passes through (:50-52)."""
    if len(corrs) == 1:
        return corrs[0], masks[0]
    s = np.stack(scores, 0)
    s = s - s.max(0, keepdims=True)
    e = np.exp(s)
    wts = (e / e.sum(0, keepdims=True)).astype(F32) + F32(1e-9)
    vw = [wts[v] * masks[v] for v in range(len(corrs))]
    wsum = np.sum(np.stack(vw, 0), 0, dtype=F32)
    fmask = (wsum != 0).astype(F32)
    csum = np.sum(np.stack([c * x for c, x in zip(corrs, vw)], 0), 0, dtype=F32)
    return (csum / (wsum + F32(1e-9)) * fmask).astype(F32), fmask


# =============================================================================================
# Path B — rows B1..B6
# =============================================================================================
def mvsnet_proj_matrices(intrinsics, poses, key_idx):
    """B1. mvsnet.py:76-91.  intrinsics[v] (3,3) pixel units, poses[v] (4,4) view_to_key... as given.
    Returns list of (4,4): K[:2]*=0.25; P[:3,:4] = K @ pose[:3,:4]; key view inverted."""
    out = []
    for v, (K, P) in enumerate(zip(intrinsics, poses)):
        K = np.asarray(K, F32) * np.array([[0.25] * 3, [0.25] * 3, [1.0] * 3], F32)
        P = np.asarray(P, F32).copy()
        P[:3, :4] = K @ P[:3, :4]
        out.append(np.linalg.inv(P).astype(F32) if v == key_idx else P)
    return out


def homo_warp_grid(src_proj, ref_proj_inv, depth_values, H, W):
    """B2 (grid part). blocks/utils.py:234-257: un-normalised sampling indices (B,D,H,W) x2."""
    B, D = depth_values.shape
    transform = np.matmul(src_proj.astype(F32), ref_proj_inv.astype(F32)).astype(F32)
    R, T = transform[:, :3, :3], transform[:, :3, 3]
    x = np.arange(W, dtype=F32)[None, None, None, :]
    y = np.arange(H, dtype=F32)[None, None, :, None]
    d = depth_values.astype(F32)[:, :, None, None]
    rr = lambda i, j: R[:, i, j][:, None, None, None]
    tt = lambda i: T[:, i][:, None, None, None]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        X = rr(0, 0) * (x * d) + rr(0, 1) * (y * d) + rr(0, 2) * (F32(1.0) * d) + tt(0)
        Y = rr(1, 0) * (x * d) + rr(1, 1) * (y * d) + rr(1, 2) * (F32(1.0) * d) + tt(1)
        Z = rr(2, 0) * (x * d) + rr(2, 1) * (y * d) + rr(2, 2) * (F32(1.0) * d) + tt(2)
        gx = (X / Z) / F32((W - 1) / 2) - F32(1.0)
        gy = (Y / Z) / F32((H - 1) / 2) - F32(1.0)
        return unnormalize(gx, W).astype(F32), unnormalize(gy, H).astype(F32)


def homo_warp(src_feat, src_proj, ref_proj_inv, depth_values):
    """B2. homo_warp, blocks/utils.py:222-268 -> (B,C,D,H,W)."""
    B, C, H, W = src_feat.shape
    ix, iy = homo_warp_grid(src_proj, ref_proj_inv, depth_values, H, W)
    return np.stack([grid_sample_zeros(src_feat[b], ix[b], iy[b]) for b in range(B)], 0)


def warp_variance(key_feat, src_feats, src_projs, ref_proj_inv, depth_values):
    """B2+B3. mvsnet.py:124-136: var = sum(x^2)/(V+1) - (sum(x)/(V+1))^2 over key + V sources."""
    B, C, H, W = key_feat.shape
    D = depth_values.shape[1]
    nv = F32(len(src_feats) + 1)
    vsum = np.repeat(key_feat[:, :, None], D, 2).astype(F32)
    vsq = vsum ** 2
    for f, P in zip(src_feats, src_projs):
        wv = homo_warp(f, P, ref_proj_inv, depth_values)
        vsum = vsum + wv
        vsq = vsq + wv ** 2
    return (vsq / nv - (vsum / nv) ** 2).astype(F32)


def conv3d(x, weight, stride=1, padding=1):
    """NCDHW cross-correlation, kernel 3x3x3."""
    N, C, D, H, W = x.shape
    O = weight.shape[0]
    k = weight.shape[2]
    p = padding
    xp = np.pad(x, ((0, 0), (0, 0), (p, p), (p, p), (p, p)))
    Do, Ho, Wo = [(s + 2 * p - k) // stride + 1 for s in (D, H, W)]
    out = np.zeros((N, O, Do, Ho, Wo), F32)
    for a in range(k):
        for b in range(k):
            for c in range(k):
                patch = xp[:, :, a:a + stride * Do:stride, b:b + stride * Ho:stride, c:c + stride * Wo:stride]
                out += np.einsum("ncdhw,oc->nodhw", patch, weight[:, :, a, b, c], optimize=True)
    return out.astype(F32)


def conv_transpose3d(x, weight, stride=2, padding=1, output_padding=1):
    """ConvTranspose3d, weight (Cin,Cout,3,3,3): out[o, i*s - p + k] += x[c, i] * w[c, o, k]."""
    N, C, D, H, W = x.shape
    O, k = weight.shape[1], weight.shape[2]
    Do, Ho, Wo = [(s - 1) * stride - 2 * padding + k + output_padding for s in (D, H, W)]
    full = np.zeros((N, O, Do + 2 * padding + k, Ho + 2 * padding + k, Wo + 2 * padding + k), F32)
    for a in range(k):
        for b in range(k):
            for c in range(k):
                contrib = np.einsum("ncdhw,co->nodhw", x, weight[:, :, a, b, c], optimize=True)
                full[:, :, a:a + stride * D:stride, b:b + stride * H:stride, c:c + stride * W:stride] += contrib
    return full[:, :, padding:padding + Do, padding:padding + Ho, padding:padding + Wo].astype(F32)


def bn_eval(x, sd, prefix, eps=1e-5):
    sh = (1, -1) + (1,) * (x.ndim - 2)
    scale = sd[prefix + "weight"] / np.sqrt(sd[prefix + "running_var"] + F32(eps))
    return ((x - sd[prefix + "running_mean"].reshape(sh)) * scale.reshape(sh) + sd[prefix + "bias"].reshape(sh)).astype(F32)


def cost_reg_net(x, sd, prefix="", return_all=False):
    """B4. CostRegNet.forward, mvsnet_components.py:111-123 (BatchNorm in eval mode)."""
    relu = lambda v: np.maximum(v, 0)

    def cbr(v, name, stride=1):
        return relu(bn_eval(conv3d(v, sd[f"{prefix}{name}.conv.weight"], stride), sd, f"{prefix}{name}.bn."))

    def dbr(v, name):
        return relu(bn_eval(conv_transpose3d(v, sd[f"{prefix}{name}.0.weight"]), sd, f"{prefix}{name}.1."))

    conv0 = cbr(x, "conv0")
    conv1 = cbr(conv0, "conv1", 2)
    conv2 = cbr(conv1, "conv2")
    conv4 = cbr(cbr(conv2, "conv3", 2), "conv4")
    y = cbr(cbr(conv4, "conv5", 2), "conv6")
    y = conv4 + dbr(y, "conv7")
    y = conv2 + dbr(y, "conv9")
    y = conv0 + dbr(y, "conv11")
    out = conv3d(y, sd[prefix + "prob.weight"]) + sd[prefix + "prob.bias"].reshape(1, -1, 1, 1, 1)
    if return_all:
        return out.astype(F32), dict(conv0=conv0, conv1=conv1)
    return out.astype(F32)


def softmax_regress(cost, depth_values):
    """B5+B6. mvsnet.py:139-160.  cost (B,D,h,w), depth_values (B,D) ->
    depth (B,h,w), confidence (B,h,w), depth_index (B,h,w) int64."""
    B, D, h, w = cost.shape
    c = cost - cost.max(1, keepdims=True)
    e = np.exp(c.astype(F32))
    p = (e / e.sum(1, keepdims=True, dtype=F32)).astype(F32)
    depth = np.sum(p * depth_values[:, :, None, None].astype(F32), 1, dtype=F32)
    fidx = np.sum(p * np.arange(D, dtype=F32)[None, :, None, None], 1, dtype=F32)
    idx = fidx.astype(np.int64)  # .long(): truncation
    pp = np.pad(p, ((0, 0), (1, 2), (0, 0), (0, 0)))
    sum4 = pp[:, 0:D] + pp[:, 1:D + 1] + pp[:, 2:D + 2] + pp[:, 3:D + 3]  # p[j-1..j+2]
    conf = np.take_along_axis(sum4, np.clip(idx, 0, D - 1)[:, None], 1)[:, 0]
    return depth.astype(F32), conf.astype(F32), idx


# ------------------------------------------------------------------------------------------------
# Input resize of the adapters
# ------------------------------------------------------------------------------------------------
def _resize_axis_table(n_in, n_out):
    """Per output index: (i0, i1, w0, w1) in float64 — scipy.ndimage.zoom's NI_ZoomShift with grid_mode=True,
    mode='mirror', order=1: c = (o + 0.5) * n_in / n_out - 0.5, mirrored into [0, n_in - 1]."""
    o = np.arange(n_out, dtype=np.float64)
    if n_in <= 1:
        z = np.zeros(n_out, np.int64)
        return z, z, np.ones(n_out), np.zeros(n_out)
    zoom = np.float64(n_in) / np.float64(n_out)
    c = (o + 0.5) * zoom - 0.5
    sz2 = 2 * n_in - 2
    c = np.where(c < 0, -c, c)
    c = np.where(c > n_in - 1, sz2 - c, c)
    st = np.floor(c)
    w1 = c - st
    i0 = st.astype(np.int64)
    i1 = i0 + 1
    i1 = np.where(i1 > n_in - 1, sz2 - i1, i1)
    return i0, i1, 1.0 - w1, w1


def resize_order1(img, ht, wd):
    """ResizeInputs' image branch (rmvd/data/transforms.py:64-66): skimage.transform.resize(image, (..., ht, wd), order=1)
    for UPSCALING.  skimage is not installed here; for this case (no anti-aliasing because no axis shrinks, float32 kept,
    clip a no-op for a convex combination, mode 'reflect' -> ndimage 'mirror') it delegates to
    scipy.ndimage.zoom(order=1, mode='mirror', grid_mode=True), which IS installed and which tests/test_resize_cpu.py pins
    this restatement against bit for bit.  img (..., H, W) float32 -> (..., ht, wd) float32."""
    H, W = img.shape[-2:]
    if ht < H or wd < W:
        raise ValueError("resize_order1 restates the upscaling case only (downscaling adds skimage's Gaussian anti-aliasing)")
    y0, y1, wy0, wy1 = _resize_axis_table(H, ht)
    x0, x1, wx0, wx1 = _resize_axis_table(W, wd)
    a = np.asarray(img, dtype=np.float32).astype(np.float64)
    r0, r1 = a[..., y0, :], a[..., y1, :]
    wy0, wy1 = wy0[:, None], wy1[:, None]
    t = (r0[..., x0] * wy0) * wx0
    t = t + (r0[..., x1] * wy0) * wx1
    t = t + (r1[..., x0] * wy1) * wx0
    t = t + (r1[..., x1] * wy1) * wx1
    return t.astype(np.float32)


def resize_inputs(images, intrinsics, ht, wd):
    """ResizeInputs.__call__ (transforms.py:56-74): images resized with order 1, intrinsics scaled by
    [[wd/orig_wd]*3, [ht/orig_ht]*3, [1]*3] as a float32 array."""
    orig_ht, orig_wd = images[0].shape[-2:]
    images = [resize_order1(im, ht, wd) for im in images]
    if intrinsics is not None:
        scale = np.array([[wd / orig_wd] * 3, [ht / orig_ht] * 3, [1.0] * 3], dtype=np.float32)
        intrinsics = [k * scale for k in intrinsics]
    return images, intrinsics


# =============================================================================================
# Backward of the sweep ops w.r.t. the feature maps.  The sampling grids carry no gradient
# (planesweep_corr.py:436,464,489 compute them under no_grad; homo_warp's grid depends on calibration only), so the
# vector-Jacobian products are the transposes of the bilinear gathers.  Pinned against autograd through the imported
# reference (tests/golden/g10_grads.npz, tests/test_oracle_golden.py).
# =============================================================================================
def grid_sample_zeros_backward(gout, ix, iy, hs, ws):
    """Transpose of grid_sample_zeros: gout (C, *ix.shape) -> gradient w.r.t. img (C,hs,ws)."""
    C = gout.shape[0]
    gimg = np.zeros((C, hs, ws), np.float64)
    for xi, yi, wgt, inb in bilinear_taps(ix, iy, hs, ws):
        w_eff = np.where(inb, wgt, F32(0.0)).astype(np.float64)
        flat = (yi * ws + xi).ravel()
        contrib = (gout.reshape(C, -1).astype(np.float64) * w_eff.ravel()[None])
        for c in range(C):
            gimg[c] += np.bincount(flat, weights=contrib[c], minlength=hs * ws).reshape(hs, ws)
    return gimg.astype(F32)


def warp_variance_backward(key_feat, src_feats, src_projs, ref_proj_inv, depth_values, gvar):
    """VJP of warp_variance (mvsnet.py:124-135 + blocks/utils.py:222-268): gvar (B,C,D,H,W) ->
    (dkey (B,C,H,W), [dsrc_v (B,C,H,W)]).  d var / d x_v = 2 (x_v - mean) / (V+1) for every volume x_v (key included)."""
    B, C, H, W = key_feat.shape
    D = depth_values.shape[1]
    nv = F32(len(src_feats) + 1)
    grids = [homo_warp_grid(P, ref_proj_inv, depth_values, H, W) for P in src_projs]
    vols = [np.repeat(key_feat[:, :, None], D, 2).astype(F32)]
    for f, (ix, iy) in zip(src_feats, grids):
        vols.append(np.stack([grid_sample_zeros(f[b], ix[b], iy[b]) for b in range(B)], 0))
    mean = sum(vols) / nv
    dkey = (gvar * F32(2.0) * (vols[0] - mean) / nv).sum(2).astype(F32)
    dsrcs = []
    for f, (ix, iy), x in zip(src_feats, grids, vols[1:]):
        gx = (gvar * F32(2.0) * (x - mean) / nv).astype(F32)
        dsrcs.append(np.stack([grid_sample_zeros_backward(gx[b], ix[b], iy[b], H, W) for b in range(B)], 0))
    return dkey, dsrcs


def sweep_corr_view_backward(feat_key, feat_src, us, vs, visible, gcorr):
    """VJP of sweep_corr_view w.r.t. the two feature maps; the 0/1 mask is a constant (planesweep_corr.py:99-104)."""
    N, C, h, w = feat_key.shape
    hs, ws = feat_src.shape[-2:]
    S = us.shape[1]
    inv_sqrt_c = F32(1.0) / np.sqrt(F32(C))
    dkey = np.zeros_like(feat_key, dtype=np.float64)
    dsrc = np.zeros_like(feat_src, dtype=np.float64)
    for n in range(N):
        for s in range(S):
            gx = F32(2.0) * us[n, s] / F32(ws) - F32(1.0)
            gy = F32(2.0) * vs[n, s] / F32(hs) - F32(1.0)
            ix, iy = unnormalize(gx, ws), unnormalize(gy, hs)
            taps = bilinear_taps(ix, iy, hs, ws)
            inb_sum = np.zeros((h, w), F32)
            for xi, yi, wgt, inb in taps:
                inb_sum += np.where(inb, wgt, F32(0.0))
            m = np.where(inb_sum < F32(0.9999), F32(0.0), F32(1.0)) * visible[n, s].astype(F32)
            coef = (gcorr[n, s] * m * inv_sqrt_c).astype(np.float64)  # (h,w)
            for xi, yi, wgt, inb in taps:
                w_eff = np.where(inb, wgt, F32(0.0)).astype(np.float64) * coef
                dkey[n] += feat_src[n][:, yi, xi] * w_eff[None]
                flat = (yi * ws + xi).ravel()
                contrib = feat_key[n].reshape(C, -1).astype(np.float64) * w_eff.ravel()[None]
                for c in range(C):
                    dsrc[n, c] += np.bincount(flat, weights=contrib[c], minlength=hs * ws).reshape(hs, ws)
    return dkey.astype(F32), dsrc.astype(F32)


def planesweep_correlation_backward(feat_key, intrinsics_key, feat_sources, source_to_key_transforms, invdepths, gcorrs,
                                    intrinsics_sources=None):
    """VJP of planesweep_correlation w.r.t. feat_key and feat_sources: -> (dkey, [dsrc_v])."""
    N, C, h, w = feat_key.shape
    if intrinsics_sources is None:
        intrinsics_sources = [intrinsics_key] * len(feat_sources)
    inv = np.asarray(invdepths, F32).reshape(np.asarray(invdepths).shape[0], -1)
    inv_n = np.broadcast_to(inv, (N, inv.shape[1]))
    dkey = np.zeros_like(feat_key)
    dsrcs = []
    for fs, T, Ks, g in zip(feat_sources, source_to_key_transforms, intrinsics_sources, gcorrs):
        co = epipolar_coeffs(intrinsics_key, Ks, T, h, w, fs.shape[2], fs.shape[3])
        us, vs, vis = sweep_grids(co, inv_n)
        dk, ds = sweep_corr_view_backward(feat_key, fs, us, vs, vis, g)
        dkey = dkey + dk
        dsrcs.append(ds)
    return dkey, dsrcs


def fuse_views_backward(corrs, masks, scores, gfused):
    """VJP of fuse_views (learned_fusion.py:32-48) w.r.t. corrs[v] and scores[v]; masks and the fused mask are constants.
    w = softmax_v(score) + 1e-9, u_v = w_v m_v, W = sum u, fused = fm * (sum c_v u_v) / (W + 1e-9)."""
    V = len(corrs)
    s = np.stack(scores, 0).astype(np.float64)
    e = np.exp(s - s.max(0, keepdims=True))
    p = e / e.sum(0, keepdims=True)  # softmax over views, (V,N,1,h,w)
    u = [(p[v] + 1e-9) * masks[v] for v in range(V)]
    W = sum(u)
    fm = (W != 0).astype(np.float64)
    num = sum(c * x for c, x in zip(corrs, u))
    den = W + 1e-9
    g = gfused.astype(np.float64) * fm
    dcorrs = [(g * u[v] / den).astype(F32) for v in range(V)]
    # d fused / d u_v = (c_v - num/den) / den ; u_v = (p_v + 1e-9) m_v ; p = softmax(score) over v, score broadcast over S
    du = [g * (corrs[v] - num / den) / den * masks[v] for v in range(V)]           # = d loss / d p_v, per (N,S,h,w)
    dp = np.stack([d.sum(1, keepdims=True) for d in du], 0)                        # scores are (N,1,h,w): sum over S
    dscore = p * (dp - (p * dp).sum(0, keepdims=True))
    return dcorrs, [dscore[v].astype(F32) for v in range(V)]


# =============================================================================================
# Other consumers of the sweep, restated from the reference's CUDA-only code
# =============================================================================================
def cvp_proj_cost(ref_feature, src_features, ref_in, src_in, ref_ex, src_ex, depth_hypos, alias_bug=True):
    """proj_cost, rmvd/models/blocks/cvp_mvsnet_components.py:375-456 (depth_hypos (B,D,H,W)), and the coarse level of
    CVPMVSNet.forward with homo_warping (:192-245; depth_hypos (B,D)).  alias_bug=True follows the reference literally:
    `volume_sum = ref_volume; volume_sq_sum = volume_sum.pow_(2)` squares the one tensor both names refer to
    (cvp_mvsnet_components.py:393-394, cvp_mvsnet.py:129-130)."""
    B, C, H, W = ref_feature.shape
    dh = np.asarray(depth_hypos, F32)
    D = dh.shape[1]
    if dh.ndim == 2:
        dh = np.broadcast_to(dh[:, :, None, None], (B, D, H, W))
    nsrc = len(src_features)
    vol = np.repeat(ref_feature[:, :, None], D, 2).astype(F32)
    vsq = vol ** 2
    vsum = vsq.copy() if alias_bug else vol
    last = np.array([[0, 0, 0, 1.0]], F32)
    x = np.arange(W, dtype=F32)[None, None, None, :]
    y = np.arange(H, dtype=F32)[None, None, :, None]
    for s in range(nsrc):
        warped = np.zeros((B, C, D, H, W), F32)
        for b in range(B):
            src_proj = np.concatenate([(src_in[b, s] @ src_ex[b, s][:3]).astype(F32), last], 0)
            ref_proj = np.concatenate([(ref_in[b] @ ref_ex[b][:3]).astype(F32), last], 0)
            proj = (src_proj @ np.linalg.inv(ref_proj)).astype(F32)
            R, T = proj[:3, :3], proj[:3, 3]
            d = dh[b][None]  # (1,D,H,W)
            with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
                # rot @ (x, y, 1), then * depth, then + trans (cvp_mvsnet_components.py:429-436)
                rx = R[0, 0] * x + R[0, 1] * y + R[0, 2]
                ry = R[1, 0] * x + R[1, 1] * y + R[1, 2]
                rz = R[2, 0] * x + R[2, 1] * y + R[2, 2]
                X, Y, Z = rx * d + T[0], ry * d + T[1], rz * d + T[2]
                gx = (X / Z) / F32((W - 1) / 2) - F32(1.0)
                gy = (Y / Z) / F32((H - 1) / 2) - F32(1.0)
                ix, iy = unnormalize(gx, W).astype(F32)[0], unnormalize(gy, H).astype(F32)[0]
            warped[b] = grid_sample_zeros(src_features[s][b], ix, iy)
        vsum = vsum + warped
        vsq = vsq + warped ** 2
    return (vsq / F32(nsrc + 1) - (vsum / F32(nsrc + 1)) ** 2).astype(F32)


def vis_homographies(left_cam, right_cam, depth_num, depth_start, depth_interval):
    """get_homographies, rmvd/models/blocks/utils.py:95-152 (inv=False): cams (n,2,4,4); depth_start / depth_interval
    (n,1,1,1) or (n,1,h,w) -> (n,d,h|1,w|1,3,3)."""
    n = left_cam.shape[0]
    d = depth_num
    R_l, R_r = left_cam[:, 0, :3, :3], right_cam[:, 0, :3, :3]
    t_l, t_r = left_cam[:, 0, :3, 3:4], right_cam[:, 0, :3, 3:4]
    K_l, K_r = left_cam[:, 1, :3, :3], right_cam[:, 1, :3, :3]
    depth = depth_start + depth_interval * np.arange(d, dtype=F32).reshape(1, d, 1, 1)
    depth = depth[..., None, None].astype(F32)
    K_l_inv = np.linalg.inv(K_l.astype(F32)).astype(F32)
    fronto = R_l[:, 2:3, :3]
    c_l = -np.swapaxes(R_l, -2, -1) @ t_l
    c_r = -np.swapaxes(R_r, -2, -1) @ t_r
    temp = ((c_r - c_l) @ fronto).reshape(n, 1, 1, 1, 3, 3)
    mm0 = np.eye(3, dtype=F32).reshape(1, 1, 1, 1, 3, 3) - temp / (depth + F32(1e-9))
    mm1 = (np.swapaxes(R_l, -2, -1) @ K_l_inv).reshape(n, 1, 1, 1, 3, 3)
    return (K_r.reshape(n, 1, 1, 1, 3, 3) @ R_r.reshape(n, 1, 1, 1, 3, 3) @ (mm0 @ mm1)).astype(F32)


def vis_cost_volumes(ref_feat, ref_cam, srcs_feat, srcs_cam, depth_num, depth_start, depth_interval, groups=8):
    """SingleStage.build_cost_volume (vis_mvsnet_singlestage.py:86-122, s_scale = d_scale = 1) + groupwise_correlation
    (blocks/utils.py:71-89) per source view: homography_warping (:176-186) maps pixel centres (x+0.5, y+0.5), divides by
    z + 1e-9, interpolate() (:163-172) normalises by the size, clamps to +-1.1 and samples with zero padding."""
    n, C, h, w = ref_feat.shape
    xs = (np.arange(w, dtype=F32) + F32(0.5))[None, :].repeat(h, 0)
    ys = (np.arange(h, dtype=F32) + F32(0.5))[:, None].repeat(w, 1)
    grid = np.stack([xs, ys, np.ones_like(xs)], -1)[..., None]  # (h,w,3,1)
    outs = []
    for sf, sc in zip(srcs_feat, srcs_cam):
        Hs = vis_homographies(ref_cam, sc, depth_num, depth_start, depth_interval)  # (n,d,h|1,w|1,3,3)
        vol = np.zeros((n, groups, depth_num, h, w), F32)
        for b in range(n):
            for k in range(depth_num):
                with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
                    wc = (Hs[b, k] @ grid)[..., 0]  # (h,w,3)
                    cx = wc[..., 0] / (wc[..., 2] + F32(1e-9))
                    cy = wc[..., 1] / (wc[..., 2] + F32(1e-9))
                    gx = np.clip(cx / F32(w) * F32(2) - F32(1), F32(-1.1), F32(1.1))
                    gy = np.clip(cy / F32(h) * F32(2) - F32(1), F32(-1.1), F32(1.1))
                    warped = grid_sample_zeros(sf[b], unnormalize(gx, w).astype(F32), unnormalize(gy, h).astype(F32))
                vol[b, :, k] = (ref_feat[b] * warped).reshape(groups, C // groups, h, w).sum(1)
        outs.append(vol)
    return outs
