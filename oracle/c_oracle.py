"""ctypes front-end of oracle/mvd_oracle_c.c (CPU ORACLE, test infrastructure only).  Same function
names and array conventions as oracle/mvd_oracle.py; used where the numpy form would be too slow
(mid-size parity cases, bench.py's cpu_baseline)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libmvd_oracle.so")
_lib = None
F32 = np.float32
_fp = ctypes.POINTER(ctypes.c_float)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `make -C oracle`")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def num_threads():
    return int(load().orc_num_threads())


def _c(a):
    return np.ascontiguousarray(a, dtype=F32)


def _p(a):
    return a.ctypes.data_as(_fp) if a is not None else None


def _pa(arrs):
    return (_fp * len(arrs))(*[_p(a) for a in arrs])


def warp_variance(key_feat, src_feats, src_projs, ref_proj_inv, depth_values, warped_only=False):
    key_feat = _c(key_feat)
    src_feats = [_c(s) for s in src_feats]
    src_projs = [_c(p) for p in src_projs]
    ref_proj_inv, depth_values = _c(ref_proj_inv), _c(depth_values)
    B, C, h, w = key_feat.shape
    D = depth_values.shape[1]
    out = np.empty((B, C, D, h, w), F32)
    load().orc_warp_variance(_p(key_feat), _pa(src_feats), _pa(src_projs), _p(ref_proj_inv), _p(depth_values),
                             B, C, D, h, w, len(src_feats), int(warped_only), _p(out))
    return out


def homo_warp(src_feat, src_proj, ref_proj_inv, depth_values):
    return warp_variance(src_feat, [src_feat], [src_proj], ref_proj_inv, depth_values, warped_only=True)


def _fold(sd, prefix, eps=1e-5):
    scale = sd[prefix + "weight"] / np.sqrt(sd[prefix + "running_var"] + F32(eps))
    return _c(scale), _c(sd[prefix + "bias"] - sd[prefix + "running_mean"] * scale)


def conv3d(x, wgt, scale, shift, stride=1, relu=True, skip=None):
    """x (Cin,D,h,w) -> (Cout,Do,ho,wo)"""
    x, wgt = _c(x), _c(wgt)
    Cin, D, h, w = x.shape
    Cout = wgt.shape[0]
    s = stride
    y = np.empty((Cout, D // s, h // s, w // s), F32)
    skip = _c(skip) if skip is not None else None
    load().orc_conv3d(_p(x), _p(wgt), _p(_c(scale)), _p(_c(shift)), _p(skip), Cin, Cout, D, h, w, s, int(relu), _p(y))
    return y


def deconv3d(x, wgt, scale, shift, relu=True, skip=None):
    x, wgt = _c(x), _c(wgt)
    Cin, D, h, w = x.shape
    Cout = wgt.shape[1]
    y = np.empty((Cout, 2 * D, 2 * h, 2 * w), F32)
    skip = _c(skip) if skip is not None else None
    load().orc_deconv3d(_p(x), _p(wgt), _p(_c(scale)), _p(_c(shift)), _p(skip), Cin, Cout, D, h, w, int(relu), _p(y))
    return y


def cost_reg_net(x, sd, prefix=""):
    """CostRegNet.forward (mvsnet_components.py:111-123), eval-mode BN folded.  x (B,32,D,h,w) -> (B,1,D,h,w)."""
    outs = []
    for xb in x:
        def cbr(v, name, stride=1):
            return conv3d(v, sd[f"{prefix}{name}.conv.weight"], *_fold(sd, f"{prefix}{name}.bn."), stride=stride)

        def dbr(v, name, skip):
            return deconv3d(v, sd[f"{prefix}{name}.0.weight"], *_fold(sd, f"{prefix}{name}.1."), skip=skip)

        conv0 = cbr(xb, "conv0")
        conv2 = cbr(cbr(conv0, "conv1", 2), "conv2")
        conv4 = cbr(cbr(conv2, "conv3", 2), "conv4")
        y = cbr(cbr(conv4, "conv5", 2), "conv6")
        y = dbr(y, "conv7", conv4)
        y = dbr(y, "conv9", conv2)
        y = dbr(y, "conv11", conv0)
        outs.append(conv3d(y, sd[prefix + "prob.weight"], np.ones(1, F32), sd[prefix + "prob.bias"], relu=False))
    return np.stack(outs, 0)


def softmax_regress(cost, depth_values):
    cost, depth_values = _c(cost), _c(depth_values)
    B, D, h, w = cost.shape
    depth = np.empty((B, h, w), F32)
    conf = np.empty((B, h, w), F32)
    load().orc_softmax_regress(_p(cost), _p(depth_values), B, D, h, w, _p(depth), _p(conf))
    return depth, conf


def sweep_corr(feat_key, feat_sources, K_key, K_sources, Ts, invdepths):
    feat_key, K_key, invdepths = _c(feat_key), _c(K_key), _c(invdepths)
    N, C, h, w = feat_key.shape
    S = invdepths.shape[1]
    corrs, masks = [], []
    for fs, Ks, T in zip(feat_sources, K_sources, Ts):
        fs, Ks, T = _c(fs), _c(Ks), _c(T)
        corr = np.empty((N, S, h, w), F32)
        mask = np.empty((N, S, h, w), F32)
        load().orc_sweep_corr(_p(feat_key), _p(fs), _p(K_key), _p(Ks), _p(T), _p(invdepths),
                              int(invdepths.shape[0] == N and N > 1), N, C, h, w, fs.shape[2], fs.shape[3], S,
                              _p(corr), _p(mask))
        corrs.append(corr)
        masks.append(mask)
    return corrs, masks


def fuse_views(corrs, masks, scores):
    corrs, masks, scores = [_c(a) for a in corrs], [_c(a) for a in masks], [_c(a) for a in scores]
    N, S, h, w = corrs[0].shape
    fused = np.empty((N, S, h, w), F32)
    fmask = np.empty((N, S, h, w), F32)
    load().orc_fuse_views(_pa(corrs), _pa(masks), _pa(scores), N, S, h, w, len(corrs), _p(fused), _p(fmask))
    return fused, fmask
