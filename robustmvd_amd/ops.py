"""Operator-level entry points of the HIP engine on torch (ROCm) tensors.  Each function validates its
arguments in Python (ValueError), allocates outputs/workspace with torch and calls one C-ABI function of
libmvd_hip.so on the tensor's device and torch's current stream.  Inference only: an input that requires grad while
autograd is recording raises (see inference_only).
"""
import functools

import torch

from . import _lib as L


def _tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _tensors(o)
    elif isinstance(obj, dict):
        for o in obj.values():
            yield from _tensors(o)


def inference_only(fn):
    """The HIP kernels behind this entry point have no backward.  The reference's ops are differentiable
    (planesweep_corr.py:514-521, learned_fusion.py:24-54), so cutting the graph silently would train a model with
    wrong gradients: with autograd recording and an input that requires grad this raises instead.  Otherwise the
    call runs under torch.no_grad()."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if torch.is_grad_enabled():
            if any(t.requires_grad for t in _tensors((args, kwargs))):
                raise RuntimeError(f"{fn.__qualname__}: an input requires grad, but this HIP path is inference-only (no "
                                   "backward is implemented); call it under torch.no_grad() or detach the inputs")
            # a bound nn.Module method (FeatureNet.forward_layout, CostRegNet.forward ...): `self` carries the parameters.
            # The reference's module is trainable; in training mode with autograd recording a silent no_grad forward would
            # hand back graph-less outputs
            owner = args[0] if args and isinstance(args[0], torch.nn.Module) else None
            if owner is not None and owner.training and any(p.requires_grad for p in owner.parameters()):
                raise RuntimeError(f"{fn.__qualname__}: the module is in training mode with trainable parameters and autograd is "
                                   "recording, but this HIP path is inference-only; call .eval() and run it under torch.no_grad()")
        with torch.no_grad():
            return fn(*args, **kwargs)
    return wrapper


def _views(ts, name, V=None):
    ts = list(ts)
    if len(ts) == 0 or len(ts) > L.MVD_MAX_VIEWS:
        raise ValueError(f"{name}: {len(ts)} views, supported 1..{L.MVD_MAX_VIEWS}")
    if V is not None and len(ts) != V:
        raise ValueError(f"{name}: {len(ts)} entries for {V} views")
    return ts


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _invdepth_mode(inv, N, h, w):
    """(1,S) shared, (N,S) per batch element, or (N,S,h,w) per key pixel (planesweep_corr.py:465-487)."""
    if inv.dim() == 2 and inv.shape[0] in (1, N):
        return L.INVDEPTH_BATCHED if (inv.shape[0] == N and N > 1) else L.INVDEPTH_SHARED
    if inv.dim() == 4 and tuple(inv.shape[0:1] + inv.shape[2:]) == (N, h, w):
        return L.INVDEPTH_PER_PIXEL
    raise ValueError(f"sampling_invdepths must be (1 or N, S) or (N, S, h, w), got {tuple(inv.shape)}")


@inference_only
def sweep_corr(feat_key, feat_sources, K_key, K_sources, T_src2key, invdepths, corr_scale=None):
    """K1. feat_key (N,C,h,w); feat_sources V x (N,C,hs,ws); K_* relative intrinsics (N,3,3);
    T_src2key V x (N,4,4); invdepths (1 or N, S) or per key pixel (N,S,h,w); corr_scale: multiplier of the dot products,
    default 1/sqrt(C) (normalize="dim").  Returns (corrs[V], masks[V]) each (N,S,h,w)."""
    lib = L.load()
    fk = L.as_f32(feat_key, "feat_key")
    if fk.dim() != 4:
        raise ValueError("feat_key must be (N,C,h,w)")
    N, C, h, w = fk.shape
    dev = fk.device
    srcs = _views(feat_sources, "feat_sources")
    V = len(srcs)
    hs, ws = srcs[0].shape[-2:]
    srcs = [L.as_f32(s, f"feat_sources[{i}]", (N, C, hs, ws), dev) for i, s in enumerate(srcs)]
    Kk = L.as_f32(K_key, "intrinsics_key", (N, 3, 3), dev)
    Ks = [L.as_f32(k, f"intrinsics_sources[{i}]", (N, 3, 3), dev) for i, k in enumerate(_views(K_sources, "intrinsics_sources", V))]
    Ts = [L.as_f32(t, f"source_to_key_transforms[{i}]", (N, 4, 4), dev) for i, t in enumerate(_views(T_src2key, "source_to_key_transforms", V))]
    inv = L.as_f32(invdepths, "sampling_invdepths", device=dev)
    mode = _invdepth_mode(inv, N, h, w)
    S = inv.shape[1]
    scale = float(corr_scale) if corr_scale is not None else 1.0 / float(C) ** 0.5
    if C % 64 != 0:
        raise ValueError(f"feature channels C={C} must be a multiple of 64")
    corrs = [torch.empty((N, S, h, w), dtype=torch.float32, device=dev) for _ in range(V)]
    masks = [torch.empty((N, S, h, w), dtype=torch.float32, device=dev) for _ in range(V)]
    wsb = lib.mvd_sweep_corr_workspace_bytes(N, C, h, w, hs, ws, V)
    wsp = _workspace(wsb, dev)
    a_src, k1 = L.ptr_array(srcs)
    a_K, k2 = L.ptr_array(Ks)
    a_T, k3 = L.ptr_array(Ts)
    a_c, k4 = L.ptr_array(corrs)
    a_m, k5 = L.ptr_array(masks)
    with torch.cuda.device(dev):
        rc = lib.mvd_sweep_corr_ex_f32(L.ptr(fk), a_src, L.ptr(Kk), a_K, a_T, L.ptr(inv), mode, scale,
                                       N, C, h, w, hs, ws, S, V, a_c, a_m, L.ptr(wsp), wsb, L.stream_of(fk))
    L.check(rc, "mvd_sweep_corr_ex_f32")
    return corrs, masks


@inference_only
def sweep_warp(feat_sources, K_key, K_sources, T_src2key, invdepths, key_size, normalize_after=False):
    """WarpOnlyCorr's sweep (planesweep_corr.py:107-140).  feat_sources V x (N,C,hs,ws); key_size (h, w) of the key feature
    map; invdepths as in sweep_corr.  Returns (warped[V] (N,S,C,h,w), masks[V] (N,S,h,w))."""
    lib = L.load()
    srcs = _views(feat_sources, "feat_sources")
    V = len(srcs)
    s0 = L.as_f32(srcs[0], "feat_sources[0]")
    if s0.dim() != 4:
        raise ValueError("feat_sources must be (N,C,hs,ws)")
    N, C, hs, ws = s0.shape
    dev = s0.device
    h, w = int(key_size[0]), int(key_size[1])
    srcs = [L.as_f32(s, f"feat_sources[{i}]", (N, C, hs, ws), dev) for i, s in enumerate(srcs)]
    Kk = L.as_f32(K_key, "intrinsics_key", (N, 3, 3), dev)
    Ks = [L.as_f32(k, f"intrinsics_sources[{i}]", (N, 3, 3), dev) for i, k in enumerate(_views(K_sources, "intrinsics_sources", V))]
    Ts = [L.as_f32(t, f"source_to_key_transforms[{i}]", (N, 4, 4), dev) for i, t in enumerate(_views(T_src2key, "source_to_key_transforms", V))]
    inv = L.as_f32(invdepths, "sampling_invdepths", device=dev)
    mode = _invdepth_mode(inv, N, h, w)
    S = inv.shape[1]
    outs = [torch.empty((N, S, C, h, w), dtype=torch.float32, device=dev) for _ in range(V)]
    masks = [torch.empty((N, S, h, w), dtype=torch.float32, device=dev) for _ in range(V)]
    a_src, k1 = L.ptr_array(srcs)
    a_K, k2 = L.ptr_array(Ks)
    a_T, k3 = L.ptr_array(Ts)
    a_o, k4 = L.ptr_array(outs)
    a_m, k5 = L.ptr_array(masks)
    with torch.cuda.device(dev):
        rc = lib.mvd_sweep_warp_f32(a_src, L.ptr(Kk), a_K, a_T, L.ptr(inv), mode, 1 if normalize_after else 0,
                                    N, C, h, w, hs, ws, S, V, a_o, a_m, L.stream_of(s0))
    L.check(rc, "mvd_sweep_warp_f32")
    return outs, masks


@inference_only
def fuse_views(corrs, masks, scores):
    """K2. corrs, masks V x (N,S,h,w); scores V x (N,1,h,w) -> fused, fused_mask (N,S,h,w)."""
    lib = L.load()
    corrs = _views(corrs, "corrs")
    V = len(corrs)
    c0 = L.as_f32(corrs[0], "corrs[0]")
    N, S, h, w = c0.shape
    dev = c0.device
    corrs = [L.as_f32(c, f"corrs[{i}]", (N, S, h, w), dev) for i, c in enumerate(corrs)]
    masks = [L.as_f32(m, f"masks[{i}]", (N, S, h, w), dev) for i, m in enumerate(_views(masks, "masks", V))]
    scores = [L.as_f32(s, f"scores[{i}]", (N, 1, h, w), dev) for i, s in enumerate(_views(scores, "scores", V))]
    fused = torch.empty_like(c0)
    fmask = torch.empty_like(c0)
    a_c, k1 = L.ptr_array(corrs)
    a_m, k2 = L.ptr_array(masks)
    a_s, k3 = L.ptr_array(scores)
    with torch.cuda.device(dev):
        rc = lib.mvd_fuse_views_f32(a_c, a_m, a_s, N, S, h, w, V, L.ptr(fused), L.ptr(fmask), L.stream_of(c0))
    L.check(rc, "mvd_fuse_views_f32")
    return fused, fmask


@inference_only
def warp_variance(key_feat, src_feats, src_projs, key_proj_inv, depth_values, channels_last=False, exact_grid=False,
                  staged=False, return_absmax=False):
    """K3. key_feat (B,C,h,w); src_feats V x (B,C,h,w); src_projs V x (B,4,4); key_proj_inv (B,4,4);
    depth_values (B,D).  Returns the variance volume (B,C,D,h,w), or (B,D,h,w,C) if channels_last.
    return_absmax: also returns max |volume| as a one-element device tensor (mvd_warp_variance_absmax_f32: a by-product of
    the store epilogue for C = 32 channel-last), which conv3d_bn_relu_split takes as its activation range.
    exact_grid: sampling positions follow the reference's operation chain rounding for rounding (MVD_GRID_EXACT).
    staged: the feature maps are the zero-bordered channel-last (B,h+3,w+3,C) copies K6 writes
    (conv2d_bn_relu(..., out_layout=LAYOUT_NHWC_BORDER)); the re-packing launches are skipped (MVD_FEAT_NHWC_BORDER)."""
    lib = L.load()
    kf = L.as_f32(key_feat, "key_feat")
    if kf.dim() != 4:
        raise ValueError("key_feat must be (B,C,h,w)" + (" / (B,h+3,w+3,C) when staged" if staged else ""))
    if staged:
        B, h, w, C = kf.shape[0], kf.shape[1] - 3, kf.shape[2] - 3, kf.shape[3]
    else:
        B, C, h, w = kf.shape
    dev = kf.device
    srcs = [L.as_f32(s, f"src_feats[{i}]", tuple(kf.shape), dev) for i, s in enumerate(_views(src_feats, "src_feats"))]
    V = len(srcs)
    projs = [L.as_f32(p, f"src_projs[{i}]", (B, 4, 4), dev) for i, p in enumerate(_views(src_projs, "src_projs", V))]
    kpi = L.as_f32(key_proj_inv, "key_proj_inv", (B, 4, 4), dev)
    dv = L.as_f32(depth_values, "depth_values", device=dev)
    if dv.dim() != 2 or dv.shape[0] != B:
        raise ValueError(f"depth_values must be (B,D), got {tuple(dv.shape)}")
    D = dv.shape[1]
    if C not in (4, 8, 16, 32, 64):
        raise ValueError(f"feature channels C={C} unsupported (4, 8, 16, 32, 64)")
    shape = (B, D, h, w, C) if channels_last else (B, C, D, h, w)
    out = torch.empty(shape, dtype=torch.float32, device=dev)
    wsb = lib.mvd_warp_variance_workspace_bytes(B, C, h, w, 0 if staged else V)
    wsp = _workspace(wsb, dev)
    a_s, k1 = L.ptr_array(srcs)
    a_p, k2 = L.ptr_array(projs)
    flags = (L.LAYOUT_NDHWC if channels_last else L.LAYOUT_NCDHW) | (L.GRID_EXACT if exact_grid else 0) | \
        (L.FEAT_NHWC_BORDER if staged else 0)
    if return_absmax:
        amax = torch.empty(1, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.mvd_warp_variance_absmax_f32(L.ptr(kf), a_s, a_p, L.ptr(kpi), L.ptr(dv), B, C, D, h, w, V, L.ptr(out),
                                                  L.ptr(amax), flags, L.ptr(wsp), wsb, L.stream_of(kf))
        L.check(rc, "mvd_warp_variance_absmax_f32")
        return out, amax
    with torch.cuda.device(dev):
        rc = lib.mvd_warp_variance_f32(L.ptr(kf), a_s, a_p, L.ptr(kpi), L.ptr(dv), B, C, D, h, w, V, L.ptr(out), flags,
                                       L.ptr(wsp), wsb, L.stream_of(kf))
    L.check(rc, "mvd_warp_variance_f32")
    return out


@inference_only
def absmax(x):
    """max |x| over a float32 device tensor as a one-element device tensor (NaNs ignored; mvd_absmax_f32, a streaming read)."""
    lib = L.load()
    x = L.as_f32(x, "x")
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.mvd_absmax_f32(L.ptr(x), x.numel(), L.ptr(out), L.stream_of(x))
    L.check(rc, "mvd_absmax_f32")
    return out


@inference_only
def to_f16(x):
    """fp32 -> fp16 (round to nearest even) through the library's converter; numel must be a multiple of 4."""
    lib = L.load()
    x = L.as_f32(x, "x")
    y = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.mvd_convert_f32_to_f16(L.ptr(x), L.ptr(y), x.numel(), L.stream_of(x))
    L.check(rc, "mvd_convert_f32_to_f16")
    return y


@inference_only
def warp_variance_f16(key_feat, src_feats, src_projs, key_proj_inv, depth_values):
    """K3, fp16-feature variant (mvd_warp_variance_f16).  key_feat, src_feats: fp16 zero-bordered channel-last maps
    (B,h+3,w+3,32); calibration fp32.  Returns the fp16 channel-last variance volume (B,D,h,w,32)."""
    lib = L.load()
    kf = L.as_f16(key_feat, "key_feat")
    if kf.dim() != 4 or kf.shape[3] != 32:
        raise ValueError("key_feat must be the fp16 zero-bordered channel-last map (B,h+3,w+3,32)")
    B, h, w = kf.shape[0], kf.shape[1] - 3, kf.shape[2] - 3
    dev = kf.device
    srcs = [L.as_f16(s, f"src_feats[{i}]", tuple(kf.shape), dev) for i, s in enumerate(_views(src_feats, "src_feats"))]
    V = len(srcs)
    projs = [L.as_f32(p, f"src_projs[{i}]", (B, 4, 4), dev) for i, p in enumerate(_views(src_projs, "src_projs", V))]
    kpi = L.as_f32(key_proj_inv, "key_proj_inv", (B, 4, 4), dev)
    dv = L.as_f32(depth_values, "depth_values", device=dev)
    if dv.dim() != 2 or dv.shape[0] != B:
        raise ValueError(f"depth_values must be (B,D), got {tuple(dv.shape)}")
    D = dv.shape[1]
    out = torch.empty((B, D, h, w, 32), dtype=torch.float16, device=dev)
    wsb = lib.mvd_warp_variance_f16_workspace_bytes(B)
    wsp = _workspace(wsb, dev)
    a_s, k1 = L.ptr_array(srcs)
    a_p, k2 = L.ptr_array(projs)
    with torch.cuda.device(dev):
        rc = lib.mvd_warp_variance_f16(L.ptr(kf), a_s, a_p, L.ptr(kpi), L.ptr(dv), B, D, h, w, V, L.ptr(out), L.ptr(wsp), wsb,
                                       L.stream_of(kf))
    L.check(rc, "mvd_warp_variance_f16")
    return out


@inference_only
def homo_warp(src_feat, src_proj, ref_proj_inv, depth_values):
    """Drop-in for rmvd.models.blocks.utils.homo_warp (blocks/utils.py:222): -> (B,C,D,H,W)."""
    lib = L.load()
    sf = L.as_f32(src_feat, "src_feat")
    if sf.dim() != 4:
        raise ValueError("src_feat must be (B,C,H,W)")
    B, C, h, w = sf.shape
    dev = sf.device
    sp = L.as_f32(src_proj, "src_proj", (B, 4, 4), dev)
    kpi = L.as_f32(ref_proj_inv, "ref_proj_inv", (B, 4, 4), dev)
    dv = L.as_f32(depth_values, "depth_values", device=dev)
    if dv.dim() != 2 or dv.shape[0] != B:
        raise ValueError(f"depth_values must be (B,D), got {tuple(dv.shape)}")
    D = dv.shape[1]
    if C not in (4, 8, 16, 32, 64):
        raise ValueError(f"feature channels C={C} unsupported (4, 8, 16, 32, 64)")
    out = torch.empty((B, C, D, h, w), dtype=torch.float32, device=dev)
    wsb = lib.mvd_warp_variance_workspace_bytes(B, C, h, w, 0)
    wsp = _workspace(wsb, dev)
    with torch.cuda.device(dev):
        rc = lib.mvd_homo_warp_f32(L.ptr(sf), L.ptr(sp), L.ptr(kpi), L.ptr(dv), B, C, D, h, w, L.ptr(out), L.ptr(wsp),
                                   wsb, L.stream_of(sf))
    L.check(rc, "mvd_homo_warp_f32")
    return out


@inference_only
def pack_conv3d_weights(weight, mode):
    """weight: Conv3d (Cout,Cin,3,3,3) or, for mode DECONV3D_STRIDE2, ConvTranspose3d (Cin,Cout,3,3,3)."""
    lib = L.load()
    wt = L.as_f32(weight, "weight")
    if wt.dim() != 5 or tuple(wt.shape[2:]) != (3, 3, 3):
        raise ValueError(f"weight must be (*,*,3,3,3), got {tuple(wt.shape)}")
    if mode == L.DECONV3D_STRIDE2:
        Cin, Cout = wt.shape[0], wt.shape[1]
    else:
        Cout, Cin = wt.shape[0], wt.shape[1]
    n = lib.mvd_conv3d_packed_weight_floats(Cin, Cout)
    if n == 0:
        raise ValueError(f"conv3d: Cin={Cin}, Cout={Cout} unsupported (Cin in 8/16/32/64, Cout in 1/8/16/32/64)")
    packed = torch.empty(n, dtype=torch.float32, device=wt.device)
    with torch.cuda.device(wt.device):
        rc = lib.mvd_pack_conv3d_weights_f32(L.ptr(wt), Cin, Cout, mode, L.ptr(packed), L.stream_of(wt))
    L.check(rc, "mvd_pack_conv3d_weights_f32")
    return packed, Cin, Cout


@inference_only
def conv3d_bn_relu(x, packed, Cin, Cout, scale, shift, mode, relu=True, skip=None, return_absmax=False):
    """K4. x (B,D,h,w,Cin) channel-last -> (B,Do,ho,wo,Cout).  return_absmax: also max |y| over the finite outputs (device,
    one float; what conv3d_bn_relu_split scales a following layer's activations by): a by-product of the store epilogue of
    the stride-2 layers, a pass over y otherwise."""
    lib = L.load()
    x = L.as_f32(x, "x")
    if x.dim() != 5 or x.shape[-1] != Cin:
        raise ValueError(f"x must be (B,D,h,w,{Cin}) channel-last, got {tuple(x.shape)}")
    B, Di, hi, wi, _ = x.shape
    dev = x.device
    if mode == L.CONV3D_STRIDE1:
        oshape = (B, Di, hi, wi, Cout)
    elif mode == L.CONV3D_STRIDE2:
        if Di % 2 or hi % 2 or wi % 2:
            raise ValueError(f"stride-2 conv needs even D,h,w, got {Di},{hi},{wi}")
        oshape = (B, Di // 2, hi // 2, wi // 2, Cout)
    elif mode == L.DECONV3D_STRIDE2:
        oshape = (B, Di * 2, hi * 2, wi * 2, Cout)
    else:
        raise ValueError(f"mode {mode}")
    scale = L.as_f32(scale, "scale", (Cout,), dev)
    shift = L.as_f32(shift, "shift", (Cout,), dev)
    if skip is not None:
        skip = L.as_f32(skip, "skip", oshape, dev)
    y = torch.empty(oshape, dtype=torch.float32, device=dev)
    if return_absmax:
        amax = torch.empty(1, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.mvd_conv3d_bn_relu_absmax_f32(L.ptr(x), L.ptr(packed), L.ptr(scale), L.ptr(shift), L.ptr(skip), L.ptr(y),
                                                   L.ptr(amax), B, Di, hi, wi, Cin, Cout, mode, int(bool(relu)), L.stream_of(x))
        L.check(rc, "mvd_conv3d_bn_relu_absmax_f32")
        return y, amax
    with torch.cuda.device(dev):
        rc = lib.mvd_conv3d_bn_relu_f32(L.ptr(x), L.ptr(packed), L.ptr(scale), L.ptr(shift), L.ptr(skip), L.ptr(y), B, Di,
                                        hi, wi, Cin, Cout, mode, int(bool(relu)), L.stream_of(x))
    L.check(rc, "mvd_conv3d_bn_relu_f32")
    return y


@inference_only
def pack_conv3d_weights_f16(weight):
    """weight: Conv3d (8,32,3,3,3) fp32 -> fp16 MFMA-fragment-ordered buffer for conv3d_bn_relu_f16in."""
    lib = L.load()
    wt = L.as_f32(weight, "weight")
    if tuple(wt.shape) != (8, 32, 3, 3, 3):
        raise ValueError(f"conv3d f16: only the 32 -> 8 first layer is built, got weight {tuple(wt.shape)}")
    packed = torch.empty(lib.mvd_conv3d_f16_packed_weight_bytes(32, 8), dtype=torch.uint8, device=wt.device)
    with torch.cuda.device(wt.device):
        rc = lib.mvd_pack_conv3d_weights_f16(L.ptr(wt), 32, 8, L.ptr(packed), L.stream_of(wt))
    L.check(rc, "mvd_pack_conv3d_weights_f16")
    return packed


@inference_only
def conv3d_bn_relu_f16in(x, packed, scale, shift, relu=True):
    """K4 first layer on fp16 MFMA: x (B,D,h,w,32) fp16 channel-last -> (B,D,h,w,8) fp32."""
    lib = L.load()
    x = L.as_f16(x, "x")
    if x.dim() != 5 or x.shape[-1] != 32:
        raise ValueError(f"x must be (B,D,h,w,32) fp16 channel-last, got {tuple(x.shape)}")
    B, D, h, w, _ = x.shape
    dev = x.device
    scale = L.as_f32(scale, "scale", (8,), dev)
    shift = L.as_f32(shift, "shift", (8,), dev)
    y = torch.empty((B, D, h, w, 8), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.mvd_conv3d_bn_relu_f16in(L.ptr(x), L.ptr(packed), L.ptr(scale), L.ptr(shift), L.ptr(y), B, D, h, w, 32, 8,
                                          int(bool(relu)), L.stream_of(x))
    L.check(rc, "mvd_conv3d_bn_relu_f16in")
    return y


@inference_only
def pack_conv3d_weights_split(weight):
    """weight: Conv3d (Cout,Cin,3,3,3) fp32, Cin 16 or 32, Cout in 8, 16, ... 64 -> per block of 8 output channels the
    [w_hi | w_lo] fp16 fragments for conv3d_bn_relu_split, followed by the channels' power-of-two scales."""
    lib = L.load()
    wt = L.as_f32(weight, "weight")
    cout, cin = (wt.shape[0], wt.shape[1]) if wt.dim() == 5 else (0, 0)
    if wt.dim() != 5 or tuple(wt.shape[2:]) != (3, 3, 3) or cin not in (16, 32) or cout % 8 or not 8 <= cout <= 64:
        raise ValueError(f"conv3d split: 16 or 32 input channels and 8..64 output channels (a multiple of 8) are built, got weight {tuple(wt.shape)}")
    packed = torch.empty(lib.mvd_conv3d_split_packed_weight_bytes(cin, cout), dtype=torch.uint8, device=wt.device)
    with torch.cuda.device(wt.device):
        rc = lib.mvd_pack_conv3d_weights_split(L.ptr(wt), cin, cout, L.ptr(packed), L.stream_of(wt))
    L.check(rc, "mvd_pack_conv3d_weights_split")
    return packed


@inference_only
def conv3d_bn_relu_split(x, packed, scale, shift, relu=True, x_absmax=None, return_absmax=False):
    """K4's stride-1 layers with 16 or 32 input channels (conv0: 32 -> 8, conv2: 16 -> 16, conv4: 32 -> 32), split-operand
    form: x (B,D,h,w,Cin) fp32 -> (B,D,h,w,Cout) fp32 (Cout = len(scale)) on fp16 MFMA with two-term operand
    splitting and power-of-two range scaling (mvd_conv3d_bn_relu_f32_split; fp32-grade results for inputs of any
    magnitude).  x_absmax: one-element device tensor with max |x| (warp_variance(..., return_absmax=True)); computed here
    with a streaming pass over x when omitted.  return_absmax: also max |y| (one-element device tensor), a by-product of the
    store epilogue."""
    lib = L.load()
    x = L.as_f32(x, "x")
    if x.dim() != 5 or x.shape[-1] not in (16, 32):
        raise ValueError(f"x must be (B,D,h,w,16 or 32) channel-last, got {tuple(x.shape)}")
    B, D, h, w, cin = x.shape
    dev = x.device
    scale = L.as_f32(scale, "scale", device=dev)
    cout = scale.numel()
    if packed.numel() != lib.mvd_conv3d_split_packed_weight_bytes(cin, cout):
        raise ValueError(f"packed weights of {packed.numel()} bytes do not belong to a {cin} -> {cout} layer")
    shift = L.as_f32(shift, "shift", (cout,), dev)
    y = torch.empty((B, D, h, w, cout), dtype=torch.float32, device=dev)
    if x_absmax is None:
        x_absmax = absmax(x)
    x_absmax = L.as_f32(x_absmax, "x_absmax", (1,), dev)
    if return_absmax:
        yam = torch.empty(1, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.mvd_conv3d_bn_relu_absmax_f32_split(L.ptr(x), L.ptr(x_absmax), L.ptr(packed), L.ptr(scale), L.ptr(shift), L.ptr(y),
                                                         L.ptr(yam), B, D, h, w, cin, cout, int(bool(relu)), L.stream_of(x))
        L.check(rc, "mvd_conv3d_bn_relu_absmax_f32_split")
        return y, yam
    with torch.cuda.device(dev):
        rc = lib.mvd_conv3d_bn_relu_f32_split(L.ptr(x), L.ptr(x_absmax), L.ptr(packed), L.ptr(scale), L.ptr(shift), L.ptr(y), B, D, h, w,
                                              cin, cout, int(bool(relu)), L.stream_of(x))
    L.check(rc, "mvd_conv3d_bn_relu_f32_split")
    return y


@inference_only
def pack_conv3d_weights_igemm(weight, mode):
    """weight: Conv3d (Cout,Cin,3,3,3), or ConvTranspose3d (Cin,Cout,3,3,3) for mode DECONV3D_STRIDE2 -> packed split-operand
    fragments for conv3d_bn_relu_igemm (mvd_pack_conv3d_weights_igemm)."""
    lib = L.load()
    wt = L.as_f32(weight, "weight")
    if wt.dim() != 5 or tuple(wt.shape[2:]) != (3, 3, 3):
        raise ValueError(f"weight must be (*,*,3,3,3), got {tuple(wt.shape)}")
    cin, cout = (wt.shape[0], wt.shape[1]) if mode == L.DECONV3D_STRIDE2 else (wt.shape[1], wt.shape[0])
    n = lib.mvd_conv3d_igemm_packed_weight_bytes(cin, cout, mode)
    if n == 0:
        raise ValueError(f"conv3d igemm: {cin} -> {cout} channels, mode {mode} is not built (Cin a multiple of 8 (16 transposed), Cout of 4)")
    packed = torch.empty(n, dtype=torch.uint8, device=wt.device)
    with torch.cuda.device(wt.device):
        rc = lib.mvd_pack_conv3d_weights_igemm(L.ptr(wt), cin, cout, mode, L.ptr(packed), L.stream_of(wt))
    L.check(rc, "mvd_pack_conv3d_weights_igemm")
    return packed


@inference_only
def conv3d_bn_relu_igemm(x, x_absmax, packed, Cin, Cout, scale, shift, mode, relu=True, skip=None, return_absmax=False, out_absmax=None):
    """K4's stride-2 / transposed / wide stride-1 layers on the split-operand implicit-GEMM kernel (mvd_conv3d_bn_relu_igemm_f32).
    x (B,D,h,w,Cin) channel-last fp32, x_absmax one-element device tensor with max |x| -> (B,Do,ho,wo,Cout); skip (like the output)
    is added after the activation.  return_absmax: also max |y| (out_absmax: a zeroed one-element device tensor to raise instead of a
    fresh one: several layers' slots can come from one zeroed buffer)."""
    lib = L.load()
    x = L.as_f32(x, "x")
    if x.dim() != 5 or x.shape[-1] != Cin:
        raise ValueError(f"x must be (B,D,h,w,{Cin}) channel-last, got {tuple(x.shape)}")
    B, Di, hi, wi, _ = x.shape
    dev = x.device
    if mode == L.CONV3D_STRIDE1:
        oshape = (B, Di, hi, wi, Cout)
    elif mode == L.CONV3D_STRIDE2:
        if Di % 2 or hi % 2 or wi % 2:
            raise ValueError(f"stride-2 conv needs even D,h,w, got {Di},{hi},{wi}")
        oshape = (B, Di // 2, hi // 2, wi // 2, Cout)
    elif mode == L.DECONV3D_STRIDE2:
        oshape = (B, Di * 2, hi * 2, wi * 2, Cout)
    else:
        raise ValueError(f"mode {mode}")
    if packed.numel() != lib.mvd_conv3d_igemm_packed_weight_bytes(Cin, Cout, mode):
        raise ValueError(f"packed weights of {packed.numel()} bytes do not belong to a {Cin} -> {Cout} layer of mode {mode}")
    scale = L.as_f32(scale, "scale", (Cout,), dev)
    shift = L.as_f32(shift, "shift", (Cout,), dev)
    if skip is not None:
        skip = L.as_f32(skip, "skip", oshape, dev)
    xam = L.as_f32(x_absmax, "x_absmax", (1,), dev)
    y = torch.empty(oshape, dtype=torch.float32, device=dev)
    yam = None
    if return_absmax:
        yam = torch.zeros(1, dtype=torch.float32, device=dev) if out_absmax is None else L.as_f32(out_absmax, "out_absmax", (1,), dev)
    wsb = lib.mvd_conv3d_igemm_workspace_bytes(B, Di, hi, wi, Cin, Cout, mode)
    wsp = _workspace(wsb, dev) if wsb else None
    with torch.cuda.device(dev):
        rc = lib.mvd_conv3d_bn_relu_igemm_f32(L.ptr(x), L.ptr(xam), L.ptr(packed), L.ptr(scale), L.ptr(shift), L.ptr(skip), L.ptr(y),
                                              L.ptr(yam), B, Di, hi, wi, Cin, Cout, mode, int(bool(relu)), L.ptr(wsp), wsb, L.stream_of(x))
    L.check(rc, "mvd_conv3d_bn_relu_igemm_f32")
    return (y, yam) if return_absmax else y


@inference_only
def pack_conv2d_weights(weight):
    """weight: Conv2d (Cout,Cin,k,k), k in (3, 5) -> (packed, Cin, Cout, k)."""
    lib = L.load()
    wt = L.as_f32(weight, "weight")
    if wt.dim() != 4 or wt.shape[2] != wt.shape[3]:
        raise ValueError(f"weight must be (Cout,Cin,k,k), got {tuple(wt.shape)}")
    Cout, Cin, k = wt.shape[0], wt.shape[1], wt.shape[2]
    n = lib.mvd_conv2d_packed_weight_floats(Cin, Cout, k)
    if n == 0:
        raise ValueError(f"conv2d: Cin={Cin}, Cout={Cout}, k={k} unsupported (Cin in 3/8/16/32, Cout in 8/16/32, k in 3/5)")
    packed = torch.empty(n, dtype=torch.float32, device=wt.device)
    with torch.cuda.device(wt.device):
        rc = lib.mvd_pack_conv2d_weights_f32(L.ptr(wt), Cin, Cout, k, L.ptr(packed), L.stream_of(wt))
    L.check(rc, "mvd_pack_conv2d_weights_f32")
    return packed, Cin, Cout, k


@inference_only
def conv2d_head(image, w0, scale0, shift0, w1, scale1, shift1, return_absmax=False):
    """FeatureNet's conv0 -> conv1 in one launch (mvd_conv2d_head_f32).  image (B,3,H,W); w0 (3,3,3,8), w1 (3,3,8,8): the Conv2d
    weights as [ky][kx][cin][cout]; folded BN scale / shift (8) per layer.  Returns (B,H,W,8) channel-last; with return_absmax also
    max |y| as a one-element device tensor (per-tile maxima from the kernel, reduced by a pass over that small array)."""
    lib = L.load()
    x = L.as_f32(image, "image")
    if x.dim() != 4 or x.shape[1] != 3:
        raise ValueError(f"image must be (B,3,H,W), got {tuple(x.shape)}")
    B, _, H, W = x.shape
    dev = x.device
    w0 = L.as_f32(w0, "w0", (3, 3, 3, 8), dev)
    w1 = L.as_f32(w1, "w1", (3, 3, 8, 8), dev)
    vs = [L.as_f32(v, n, (8,), dev) for v, n in ((scale0, "scale0"), (shift0, "shift0"), (scale1, "scale1"), (shift1, "shift1"))]
    y = torch.empty((B, H, W, 8), dtype=torch.float32, device=dev)
    tiles = torch.empty(lib.mvd_conv2d_head_tile_count(B, H, W), dtype=torch.float32, device=dev) if return_absmax else None
    with torch.cuda.device(dev):
        rc = lib.mvd_conv2d_head_f32(L.ptr(x), L.ptr(w0), L.ptr(vs[0]), L.ptr(vs[1]), L.ptr(w1), L.ptr(vs[2]), L.ptr(vs[3]), L.ptr(y),
                                     L.ptr(tiles), B, H, W, L.stream_of(x))
    L.check(rc, "mvd_conv2d_head_f32")
    return (y, absmax(tiles)) if return_absmax else y


def conv2d_bn_relu(x, packed, Cin, Cout, ksize, stride, scale, shift, relu=True, out_layout=L.LAYOUT_NHWC, out=None, out_absmax=None):
    """K6. x: (B,3,H,W) image when Cin == 3, else channel-last (B,h,w,Cin).  Returns (B,ho,wo,Cout) for LAYOUT_NHWC,
    (B,Cout,ho,wo) for LAYOUT_NCHW, or the zero-bordered (B,ho+3,wo+3,Cout) staging map for LAYOUT_NHWC_BORDER
    (`out` may pass a buffer whose border is already zero; only the interior is written).  out_absmax: a zeroed one-element
    device tensor that is raised to max |y| (mvd_conv2d_bn_relu_absmax_f32), for a split-operand layer behind this one."""
    lib = L.load()
    x = L.as_f32(x, "x")
    if Cin == 3:
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"x must be (B,3,H,W), got {tuple(x.shape)}")
        B, _, hi, wi = x.shape
        in_layout = L.LAYOUT_NCHW
    else:
        if x.dim() != 4 or x.shape[-1] != Cin:
            raise ValueError(f"x must be (B,h,w,{Cin}) channel-last, got {tuple(x.shape)}")
        B, hi, wi, _ = x.shape
        in_layout = L.LAYOUT_NHWC
    if (ksize, stride) not in ((3, 1), (5, 2)):
        raise ValueError(f"conv2d: kernel {ksize} stride {stride} unsupported (3/1 or 5/2)")
    dev = x.device
    ho, wo = (hi - 1) // stride + 1, (wi - 1) // stride + 1
    oshape = {L.LAYOUT_NHWC: (B, ho, wo, Cout), L.LAYOUT_NCHW: (B, Cout, ho, wo),
              L.LAYOUT_NHWC_BORDER: (B, ho + 3, wo + 3, Cout)}.get(out_layout)
    if oshape is None:
        raise ValueError(f"out_layout {out_layout}")
    scale = L.as_f32(scale, "scale", (Cout,), dev)
    shift = L.as_f32(shift, "shift", (Cout,), dev)
    if out is not None:
        y = L.as_f32(out, "out", oshape, dev)
        if y.data_ptr() != out.data_ptr():
            raise ValueError("out must be a contiguous fp32 tensor")
    elif out_layout == L.LAYOUT_NHWC_BORDER:
        y = torch.zeros(oshape, dtype=torch.float32, device=dev)
    else:
        y = torch.empty(oshape, dtype=torch.float32, device=dev)
    if out_absmax is not None:
        yam = L.as_f32(out_absmax, "out_absmax", (1,), dev)
        with torch.cuda.device(dev):
            rc = lib.mvd_conv2d_bn_relu_absmax_f32(L.ptr(x), in_layout, L.ptr(packed), L.ptr(scale), L.ptr(shift), L.ptr(y), L.ptr(yam),
                                                   out_layout, B, hi, wi, Cin, Cout, ksize, stride, int(bool(relu)), L.stream_of(x))
        L.check(rc, "mvd_conv2d_bn_relu_absmax_f32")
        return y
    with torch.cuda.device(dev):
        rc = lib.mvd_conv2d_bn_relu_f32(L.ptr(x), in_layout, L.ptr(packed), L.ptr(scale), L.ptr(shift), L.ptr(y), out_layout,
                                        B, hi, wi, Cin, Cout, ksize, stride, int(bool(relu)), L.stream_of(x))
    L.check(rc, "mvd_conv2d_bn_relu_f32")
    return y


@inference_only
def softmax_regress(cost, depth_values, with_confidence=True):
    """K5. cost (B,D,h,w); depth_values (B,D) -> depth (B,h,w), confidence (B,h,w) or None."""
    lib = L.load()
    c = L.as_f32(cost, "cost")
    if c.dim() != 4:
        raise ValueError("cost must be (B,D,h,w)")
    B, D, h, w = c.shape
    dv = L.as_f32(depth_values, "depth_values", (B, D), c.device)
    depth = torch.empty((B, h, w), dtype=torch.float32, device=c.device)
    conf = torch.empty((B, h, w), dtype=torch.float32, device=c.device) if with_confidence else None
    with torch.cuda.device(c.device):
        rc = lib.mvd_softmax_regress_f32(L.ptr(c), L.ptr(dv), B, D, h, w, L.ptr(depth), L.ptr(conf), L.stream_of(c))
    L.check(rc, "mvd_softmax_regress_f32")
    return depth, conf


class SplitConv2dWeights:
    """Packed split-operand weights of one 2-D layer (pack_conv2d_weights_split) with what conv2d_split needs to call it."""
    __slots__ = ("packed", "bias", "cin", "cin_pad", "cout", "kh", "kw", "stride", "mode")

    def __init__(self, packed, bias, cin, cin_pad, cout, kh, kw, stride, mode):
        self.packed, self.bias, self.cin, self.cin_pad, self.cout = packed, bias, cin, cin_pad, cout
        self.kh, self.kw, self.stride, self.mode = kh, kw, stride, mode


@inference_only
def pack_conv2d_weights_split(weight, bias=None, stride=1, mode=L.CONV2D, cin_pad=None):
    """weight: Conv2d (Cout,Cin,k,k) (mode CONV2D / CONV2D_IMAGE) or ConvTranspose2d (Cin,Cout,4,4) (mode DECONV2D), fp32 ->
    SplitConv2dWeights for conv2d_split.  cin_pad: the channel count of the (zero-padded) input slice the layer will read, a
    multiple of 8 (default: Cin rounded up to 8)."""
    lib = L.load()
    wt = L.as_f32(weight, "weight")
    if wt.dim() != 4:
        raise ValueError(f"weight must be 4-D, got {tuple(wt.shape)}")
    if mode == L.DECONV2D:
        cin, cout = wt.shape[0], wt.shape[1]
    else:
        cout, cin = wt.shape[0], wt.shape[1]
    kh, kw = wt.shape[2], wt.shape[3]
    cin_pad = (cin + 7) // 8 * 8 if cin_pad is None else int(cin_pad)
    nbytes = lib.mvd_conv2d_split_packed_weight_bytes(cin_pad, cout, kh, kw, stride, mode) if cin_pad >= cin else 0
    if nbytes == 0:
        raise ValueError(f"conv2d split: a {kh}x{kw} stride-{stride} layer (mode {mode}) with {cin} (padded {cin_pad}) -> {cout} channels is not built")
    packed = torch.empty(nbytes, dtype=torch.uint8, device=wt.device)
    with torch.cuda.device(wt.device):
        rc = lib.mvd_pack_conv2d_weights_split(L.ptr(wt), cin, cin_pad, cout, kh, kw, stride, mode, L.ptr(packed), L.stream_of(wt))
    L.check(rc, "mvd_pack_conv2d_weights_split")
    b = None if bias is None else L.as_f32(bias.detach(), "bias", (cout,), wt.device).clone()
    return SplitConv2dWeights(packed, b, cin, cin_pad, cout, kh, kw, stride, mode)


def _nhwc_slice(t, name, channels=None, free_rows=False):
    """(B,H,W,C) fp32 device view with unit channel stride -> (B,H,W,C, pixel stride, row stride, image stride) in floats.  Without
    free_rows the view must be a channel slice of a dense NHWC buffer (rows and images follow each other)."""
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 4):
        raise ValueError(f"{name} must be a float32 device tensor (B,H,W,C)")
    B, H, W, C = t.shape
    ps = t.stride(2) if W > 1 else max(C, 1)
    rs = t.stride(1) if H > 1 else W * ps
    ims = t.stride(0) if B > 1 else H * rs
    if (C > 1 and t.stride(3) != 1) or ps < C or rs < W * ps or ims < H * rs:
        raise ValueError(f"{name}: a channel slice of a channel-last buffer is needed, got shape {tuple(t.shape)} strides {t.stride()}")
    if not free_rows and (rs != W * ps or ims != H * rs):
        raise ValueError(f"{name}: a channel slice of a DENSE channel-last buffer is needed, got shape {tuple(t.shape)} strides {t.stride()}")
    if channels is not None and C != channels:
        raise ValueError(f"{name}: {C} channels, the layer takes {channels}")
    return B, H, W, C, ps, rs, ims


@inference_only
def conv2d_split(x, x_absmax, wts, act=1, slope=0.2, out=None, out_absmax=None, use_workspace=True, planar_out=False):
    """One layer of Path A's 2-D CNN on the split-operand kernel (mvd_conv2d_split_f32).  x: (B,Hi,Wi,Cin_pad) NHWC fp32, possibly
    a channel slice of a wider buffer (mode CONV2D_IMAGE: the planar (B,3,Hi,Wi) image); x_absmax: one-element device tensor with
    max |x|; wts: SplitConv2dWeights.  out: NHWC destination view (B,Ho,Wo,Cout), e.g. a slice of a concat buffer or the interior
    of a zero-bordered map (allocated if None); planar_out: allocate and return (B,Cout,Ho,Wo) instead.  out_absmax: one-element
    device tensor that receives max |out| by atomic maximum (zero it first), or None.  act 0 none / 1 LeakyReLU(slope) / 2 ReLU."""
    lib = L.load()
    if wts.mode == L.CONV2D_IMAGE:
        xi = L.as_f32(x, "x")
        if xi.dim() != 4 or xi.shape[1] != 3:
            raise ValueError(f"x must be the planar image (B,3,H,W), got {tuple(xi.shape)}")
        B, _, Hi, Wi = xi.shape
        x, xs = xi, 0
    else:
        B, Hi, Wi, _, xs, _, _ = _nhwc_slice(x, "x", wts.cin_pad)
    if wts.mode == L.DECONV2D:
        Ho, Wo = 2 * Hi, 2 * Wi
    else:
        Ho = (Hi + 2 * (wts.kh // 2) - wts.kh) // wts.stride + 1
        Wo = (Wi + 2 * (wts.kw // 2) - wts.kw) // wts.stride + 1
    dev = x.device
    if planar_out:
        if out is not None:
            raise ValueError("planar_out allocates its own output")
        out = torch.empty((B, wts.cout, Ho, Wo), dtype=torch.float32, device=dev)
        ys, rs, ims, cs = 1, Wo, wts.cout * Ho * Wo, Ho * Wo
    else:
        if out is None:
            out = torch.empty((B, Ho, Wo, wts.cout), dtype=torch.float32, device=dev)
        ob, oh, ow, _, ys, rs, ims = _nhwc_slice(out, "out", wts.cout, free_rows=True)
        cs = 1
        if (ob, oh, ow) != (B, Ho, Wo):
            raise ValueError(f"out is {tuple(out.shape)}, the layer writes ({B},{Ho},{Wo},{wts.cout})")
    xam = L.as_f32(x_absmax, "x_absmax", (1,), dev)
    yam = None if out_absmax is None else L.as_f32(out_absmax, "out_absmax", (1,), dev)
    wsb = lib.mvd_conv2d_split_workspace_bytes(B, Hi, Wi, wts.cin_pad, wts.cout, wts.kh, wts.kw, wts.stride, wts.mode) if use_workspace else 0
    wsp = _workspace(wsb, dev) if wsb else None
    with torch.cuda.device(dev):
        rc = lib.mvd_conv2d_split_f32(L.ptr(x), L.ptr(xam), L.ptr(wts.packed), L.ptr(wts.bias), L.ptr(out), L.ptr(yam), B, Hi, Wi,
                                      wts.cin_pad, xs, wts.cout, ys, rs, ims, cs, wts.kh, wts.kw, wts.stride, wts.mode, int(act),
                                      float(slope), L.ptr(wsp), wsb, L.stream_of(x))
    L.check(rc, "mvd_conv2d_split_f32")
    return out


@inference_only
def upsample2x_into(x, out, out_absmax=None):
    """F.interpolate(x, size=(2h,2w), mode="bilinear", align_corners=False) of a planar (B,C,h,w) map written into the channel-last
    slice out (B,2h,2w,C) (mvd_upsample2x_nhwc_f32: the decoder's up-sampled prediction inside the next level's concat buffer)."""
    lib = L.load()
    x = L.as_f32(x, "x")
    B, C, h, w = x.shape
    ob, oh, ow, _, ys, _, _ = _nhwc_slice(out, "out", C)
    if (ob, oh, ow) != (B, 2 * h, 2 * w):
        raise ValueError(f"out is {tuple(out.shape)}, expected ({B},{2 * h},{2 * w},{C})")
    yam = None if out_absmax is None else L.as_f32(out_absmax, "out_absmax", (1,), x.device)
    with torch.cuda.device(x.device):
        rc = lib.mvd_upsample2x_nhwc_f32(L.ptr(x), L.ptr(out), L.ptr(yam), B, C, h, w, ys, L.stream_of(x))
    L.check(rc, "mvd_upsample2x_nhwc_f32")
    return out


@inference_only
def sweep_corr_nhwc(feat_key, feat_sources, K_key, K_sources, T_src2key, invdepths, corrs, masks, corr_scale=None, corr_absmax=None):
    """K1 on its working layouts (mvd_sweep_corr_nhwc_f32): feat_key (N,h,w,C) channel-last; feat_sources V x zero-bordered
    channel-last (N,hs+3,ws+3,C); corrs, masks: V x pixel-major destinations (N,h,w,S) (channel slices allowed), filled in place.
    corr_absmax: one-element device tensor raised to max |corr| over all views (zero it first), or None."""
    lib = L.load()
    fk = L.as_f32(feat_key, "feat_key")
    N, h, w, C = fk.shape
    dev = fk.device
    srcs = _views(feat_sources, "feat_sources")
    V = len(srcs)
    hs, ws = srcs[0].shape[1] - 3, srcs[0].shape[2] - 3
    srcs = [L.as_f32(s, f"feat_sources[{i}]", (N, hs + 3, ws + 3, C), dev) for i, s in enumerate(srcs)]
    Kk = L.as_f32(K_key, "intrinsics_key", (N, 3, 3), dev)
    Ks = [L.as_f32(k, f"intrinsics_sources[{i}]", (N, 3, 3), dev) for i, k in enumerate(_views(K_sources, "intrinsics_sources", V))]
    Ts = [L.as_f32(t, f"source_to_key_transforms[{i}]", (N, 4, 4), dev) for i, t in enumerate(_views(T_src2key, "source_to_key_transforms", V))]
    inv = L.as_f32(invdepths, "sampling_invdepths", device=dev)
    mode = _invdepth_mode(inv, N, h, w)
    S = inv.shape[1]
    if C % 64 != 0:
        raise ValueError(f"feature channels C={C} must be a multiple of 64")
    ps = None
    for name, ts in (("corrs", corrs), ("masks", masks)):
        if len(ts) != V:
            raise ValueError(f"{name}: {len(ts)} entries for {V} views")
        for t in ts:
            b_, h_, w_, _, p_, _, _ = _nhwc_slice(t, name, S)
            if (b_, h_, w_) != (N, h, w) or (ps is not None and p_ != ps):
                raise ValueError(f"{name}: (N,h,w,S) = ({N},{h},{w},{S}) maps with one common pixel stride are needed")
            ps = p_
    scale = float(corr_scale) if corr_scale is not None else 1.0 / float(C) ** 0.5
    a_src, k1 = L.ptr_array(srcs)
    a_K, k2 = L.ptr_array(Ks)
    a_T, k3 = L.ptr_array(Ts)
    a_c, k4 = L.ptr_array(list(corrs))
    a_m, k5 = L.ptr_array(list(masks))
    cam = None if corr_absmax is None else L.as_f32(corr_absmax, "corr_absmax", (1,), dev)
    with torch.cuda.device(dev):
        rc = lib.mvd_sweep_corr_nhwc_f32(L.ptr(fk), a_src, L.ptr(Kk), a_K, a_T, L.ptr(inv), mode, scale, N, C, h, w, hs, ws, S, V,
                                         a_c, a_m, ps, L.ptr(cam), L.stream_of(fk))
    L.check(rc, "mvd_sweep_corr_nhwc_f32")
    return corrs, masks


@inference_only
def fuse_views_nhwc(corrs, masks, scores, out, out_absmax=None):
    """K2 on pixel-major volumes: corrs, masks V x (N,h,w,S); scores V x (N,h,w,1) or (N,1,h,w); out: (N,h,w,S) destination view
    (a channel slice of the cost-volume encoder's input buffer).  Returns out."""
    lib = L.load()
    V = len(corrs)
    N, h, w, S, ps, _, _ = _nhwc_slice(corrs[0], "corrs[0]")
    dev = corrs[0].device
    for name, ts in (("corrs", corrs), ("masks", masks)):
        for t in ts:
            if _nhwc_slice(t, name, S)[:5] != (N, h, w, S, ps):
                raise ValueError(f"{name}: equal (N,h,w,S) maps are needed")
    scores = [L.as_f32(s_.reshape(N, h * w), f"scores[{i}]", (N, h * w), dev) for i, s_ in enumerate(_views(scores, "scores", V))]
    ob, oh, ow, _, ops_, _, _ = _nhwc_slice(out, "out", S)
    if (ob, oh, ow) != (N, h, w):
        raise ValueError(f"out is {tuple(out.shape)}, expected ({N},{h},{w},{S})")
    yam = None if out_absmax is None else L.as_f32(out_absmax, "out_absmax", (1,), dev)
    a_c, k1 = L.ptr_array(list(corrs))
    a_m, k2 = L.ptr_array(list(masks))
    a_s, k3 = L.ptr_array(scores)
    with torch.cuda.device(dev):
        rc = lib.mvd_fuse_views_nhwc_f32(a_c, a_m, a_s, N, S, h, w, V, ps, L.ptr(out), None, ops_, L.ptr(yam), L.stream_of(corrs[0]))
    L.check(rc, "mvd_fuse_views_nhwc_f32")
    return out


@inference_only
def bias_leaky_relu_(x, bias, slope=0.2):
    """In place: x (N,C,H,W) contiguous <- leaky_relu(x + bias[c], slope).  Returns x."""
    lib = L.load()
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() >= 2):
        raise ValueError("bias_leaky_relu_: x must be a contiguous fp32 device tensor (N,C,...)")
    N, C = x.shape[0], x.shape[1]
    b = L.as_f32(bias, "bias", (C,), x.device)
    with torch.cuda.device(x.device):
        rc = lib.mvd_bias_leaky_relu_f32(L.ptr(x), L.ptr(b), N, C, x.numel() // (N * C), float(slope), L.stream_of(x))
    L.check(rc, "mvd_bias_leaky_relu_f32")
    return x


@inference_only
def resize_order1(images, ht, wd):
    """images (..., H, W) fp32 device tensor -> (..., ht, wd): skimage.transform.resize(order=1) for upscaling
    (rmvd/data/transforms.py:64-66), on the device (mvd_resize_order1_f32)."""
    lib = L.load()
    x = L.as_f32(images, "images")
    if x.dim() < 2:
        raise ValueError("images must be (..., H, W)")
    hi, wi = x.shape[-2:]
    if ht < hi or wd < wi:
        raise ValueError(f"resize_order1: only upscaling is built ({hi}x{wi} -> {ht}x{wd})")
    planes = x.numel() // (hi * wi)
    y = torch.empty(tuple(x.shape[:-2]) + (ht, wd), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.mvd_resize_order1_f32(L.ptr(x), L.ptr(y), planes, hi, wi, ht, wd, L.stream_of(x))
    L.check(rc, "mvd_resize_order1_f32")
    return y


@inference_only
def dispnet_head(x):
    """x (N,2,h,w) raw output of a pred_k convolution -> (pred (N,2,h,w) = [relu(x0), sigmoid(0.2 x1) * 20 - 10],
    entropy (N,1,h,w) = log(2 exp(pred1) + 1e-4) + 1) in one launch (dispnet_decoder.py:17-22,126-138)."""
    lib = L.load()
    x = L.as_f32(x, "x")
    if x.dim() != 4 or x.shape[1] != 2:
        raise ValueError(f"x must be (N,2,h,w), got {tuple(x.shape)}")
    N, _, h, w = x.shape
    pred = torch.empty_like(x)
    ent = torch.empty((N, 1, h, w), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.mvd_dispnet_head_f32(L.ptr(x), L.ptr(pred), L.ptr(ent), N, h * w, L.stream_of(x))
    L.check(rc, "mvd_dispnet_head_f32")
    return pred, ent


@inference_only
def to_channels_last_3d(x):
    """(B,C,D,h,w) -> (B,D,h,w,C) through the library's tiled transpose."""
    lib = L.load()
    x = L.as_f32(x, "x")
    B, C, D, h, w = x.shape
    y = torch.empty((B, D, h, w, C), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.mvd_nchw_to_nhwc_f32(L.ptr(x), L.ptr(y), B, C, D * h * w, L.stream_of(x))
    L.check(rc, "mvd_nchw_to_nhwc_f32")
    return y


@inference_only
def from_channels_last_3d(y):
    """(B,D,h,w,C) -> (B,C,D,h,w)."""
    lib = L.load()
    y = L.as_f32(y, "y")
    B, D, h, w, C = y.shape
    x = torch.empty((B, C, D, h, w), dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        rc = lib.mvd_nhwc_to_nchw_f32(L.ptr(y), L.ptr(x), B, C, D * h * w, L.stream_of(y))
    L.check(rc, "mvd_nhwc_to_nchw_f32")
    return x


# ------------------------------------------------------------------------------------------------
# Differentiable forms: torch.autograd.Function wrappers whose backward runs the engine's VJP
# kernels (include/mvd.h "Backward of the sweep operators").  Gradients flow to the FEATURE MAPS (and, for the fusion, to
# the score maps); calibration, depth samples and masks are constants, exactly as in the reference
# (planesweep_corr.py:436,464,489 compute the grids under no_grad).
# ------------------------------------------------------------------------------------------------
def _bordered_channels_last(x):
    """(N,C,h,w) -> zero-bordered channel-last (N,h+3,w+3,C) with the map at (1,1): the layout the kernels gather from."""
    n, c, h, w = x.shape
    out = torch.zeros((n, h + 3, w + 3, c), dtype=torch.float32, device=x.device)
    out[:, 1:h + 1, 1:w + 1] = x.permute(0, 2, 3, 1)
    return out


def _interior_nchw(g, h, w):
    return g[:, 1:h + 1, 1:w + 1].permute(0, 3, 1, 2).contiguous()


class _WarpVariance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, key_proj_inv, depth_values, n_views, key_feat, *rest):
        srcs, projs = list(rest[:n_views]), list(rest[n_views:])
        ctx.save_for_backward(key_proj_inv, depth_values, key_feat, *srcs, *projs)
        ctx.n_views = n_views
        return warp_variance(key_feat.detach(), [s.detach() for s in srcs], projs, key_proj_inv, depth_values)

    @staticmethod
    def backward(ctx, gvar):
        lib = L.load()
        V = ctx.n_views
        kpi, dv, key = ctx.saved_tensors[:3]
        srcs, projs = list(ctx.saved_tensors[3:3 + V]), list(ctx.saved_tensors[3 + V:])
        B, C, h, w = key.shape
        D = dv.shape[1]
        dev = key.device
        with torch.no_grad():
            kb = _bordered_channels_last(key.float())
            sb = [_bordered_channels_last(s.float()) for s in srcs]
            g = gvar.float().permute(0, 2, 3, 4, 1).contiguous()  # (B,D,h,w,C)
            gk = torch.empty_like(kb)
            gs = [torch.empty_like(kb) for _ in range(V)]
            pr = [L.as_f32(p, "src_proj", (B, 4, 4), dev) for p in projs]
            wsb = lib.mvd_warp_variance_backward_workspace_bytes(B)
            wsp = _workspace(wsb, dev)
            a_s, k1 = L.ptr_array(sb)
            a_p, k2 = L.ptr_array(pr)
            a_g, k3 = L.ptr_array(gs)
            with torch.cuda.device(dev):
                rc = lib.mvd_warp_variance_backward_f32(L.ptr(kb), a_s, a_p, L.ptr(L.as_f32(kpi, "key_proj_inv", (B, 4, 4), dev)),
                                                        L.ptr(L.as_f32(dv, "depth_values", (B, D), dev)), L.ptr(g), B, C, D, h, w,
                                                        V, L.ptr(gk), a_g, L.ptr(wsp), wsb, L.stream_of(kb))
            L.check(rc, "mvd_warp_variance_backward_f32")
            out = [None, None, None, _interior_nchw(gk, h, w)] + [_interior_nchw(x, h, w) for x in gs] + [None] * V
        return tuple(out)


def warp_variance_autograd(key_feat, src_feats, src_projs, key_proj_inv, depth_values):
    """Differentiable K3: like warp_variance (reference layout (B,C,D,h,w)), with gradients to key_feat and src_feats."""
    srcs = _views(src_feats, "src_feats")
    return _WarpVariance.apply(key_proj_inv, depth_values, len(srcs), key_feat, *srcs, *_views(src_projs, "src_projs", len(srcs)))


class _SweepCorr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, K_key, invdepths, n_views, corr_scale, feat_key, *rest):
        V = n_views
        srcs, Ks, Ts = list(rest[:V]), list(rest[V:2 * V]), list(rest[2 * V:])
        corrs, masks = sweep_corr(feat_key.detach(), [s.detach() for s in srcs], K_key, Ks, Ts, invdepths, corr_scale)
        ctx.save_for_backward(K_key, invdepths, feat_key, *srcs, *Ks, *Ts)
        ctx.n_views = V
        ctx.corr_scale = corr_scale
        ctx.mark_non_differentiable(*masks)
        return tuple(corrs) + tuple(masks)

    @staticmethod
    def backward(ctx, *grads):
        lib = L.load()
        V = ctx.n_views
        Kk, inv, fk = ctx.saved_tensors[:3]
        srcs = list(ctx.saved_tensors[3:3 + V])
        Ks, Ts = list(ctx.saved_tensors[3 + V:3 + 2 * V]), list(ctx.saved_tensors[3 + 2 * V:])
        N, C, h, w = fk.shape
        hs, ws = srcs[0].shape[-2:]
        S = inv.shape[1]
        dev = fk.device
        with torch.no_grad():
            key = fk.float().permute(0, 2, 3, 1).contiguous()
            sb = [_bordered_channels_last(s.float()) for s in srcs]
            gc = [(g if g is not None else torch.zeros((N, S, h, w), device=dev)).float().contiguous() for g in grads[:V]]
            gk = torch.empty_like(key)
            gs = [torch.empty_like(sb[0]) for _ in range(V)]
            Ksf = [L.as_f32(k, "K_src", (N, 3, 3), dev) for k in Ks]
            Tsf = [L.as_f32(t, "T", (N, 4, 4), dev) for t in Ts]
            a_s, k1 = L.ptr_array(sb)
            a_K, k2 = L.ptr_array(Ksf)
            a_T, k3 = L.ptr_array(Tsf)
            a_gc, k4 = L.ptr_array(gc)
            a_gs, k5 = L.ptr_array(gs)
            invf = L.as_f32(inv, "invdepths", device=dev)
            scale = float(ctx.corr_scale) if ctx.corr_scale is not None else 1.0 / float(C) ** 0.5
            with torch.cuda.device(dev):
                rc = lib.mvd_sweep_corr_backward_f32(L.ptr(key), a_s, L.ptr(L.as_f32(Kk, "K_key", (N, 3, 3), dev)), a_K, a_T,
                                                     L.ptr(invf), _invdepth_mode(invf, N, h, w), scale, a_gc, N, C, h, w, hs, ws,
                                                     S, V, L.ptr(gk), a_gs, L.stream_of(key))
            L.check(rc, "mvd_sweep_corr_backward_f32")
            out = [None, None, None, None, gk.permute(0, 3, 1, 2).contiguous()] + [_interior_nchw(x, hs, ws) for x in gs] + [None] * (2 * V)
        return tuple(out)


def sweep_corr_autograd(feat_key, feat_sources, K_key, K_sources, T_src2key, invdepths, corr_scale=None):
    """Differentiable K1: returns (corrs[V], masks[V]); gradients to feat_key and feat_sources."""
    srcs = _views(feat_sources, "feat_sources")
    V = len(srcs)
    outs = _SweepCorr.apply(K_key, invdepths, V, corr_scale, feat_key, *srcs, *_views(K_sources, "intrinsics_sources", V),
                            *_views(T_src2key, "source_to_key_transforms", V))
    return list(outs[:V]), list(outs[V:])


class _FuseViews(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n_views, *rest):
        V = n_views
        corrs, masks, scores = list(rest[:V]), list(rest[V:2 * V]), list(rest[2 * V:])
        fused, fmask = fuse_views([c.detach() for c in corrs], [m.detach() for m in masks], [s.detach() for s in scores])
        ctx.save_for_backward(*corrs, *masks, *scores)
        ctx.n_views = V
        ctx.mark_non_differentiable(fmask)
        return fused, fmask

    @staticmethod
    def backward(ctx, gfused, _gmask):
        lib = L.load()
        V = ctx.n_views
        t = ctx.saved_tensors
        corrs, masks, scores = list(t[:V]), list(t[V:2 * V]), list(t[2 * V:])
        N, S, h, w = corrs[0].shape
        dev = corrs[0].device
        with torch.no_grad():
            cs = [L.as_f32(c, "corr", (N, S, h, w), dev) for c in corrs]
            ms = [L.as_f32(m, "mask", (N, S, h, w), dev) for m in masks]
            ss = [L.as_f32(s, "score", (N, 1, h, w), dev) for s in scores]
            g = gfused.float().contiguous()
            gcs = [torch.empty_like(cs[0]) for _ in range(V)]
            gss = [torch.empty_like(ss[0]) for _ in range(V)]
            a_c, k1 = L.ptr_array(cs)
            a_m, k2 = L.ptr_array(ms)
            a_s, k3 = L.ptr_array(ss)
            a_gc, k4 = L.ptr_array(gcs)
            a_gs, k5 = L.ptr_array(gss)
            with torch.cuda.device(dev):
                rc = lib.mvd_fuse_views_backward_f32(a_c, a_m, a_s, L.ptr(g), N, S, h, w, V, a_gc, a_gs, L.stream_of(g))
            L.check(rc, "mvd_fuse_views_backward_f32")
        return (None,) + tuple(gcs) + (None,) * V + tuple(gss)


def fuse_views_autograd(corrs, masks, scores):
    """Differentiable K2: gradients to corrs and scores."""
    corrs = _views(corrs, "corrs")
    V = len(corrs)
    return _FuseViews.apply(V, *corrs, *_views(masks, "masks", V), *_views(scores, "scores", V))


def needs_grad(*objs):
    return torch.is_grad_enabled() and any(t.requires_grad for t in _tensors(objs))
