"""End-to-end CPU ORACLE forward passes (test infrastructure only): the hot-path rows run on the C/OpenMP
oracle (oracle/mvd_oracle_c.c), the adjacent 2-D CNN layers — which the product also leaves to a vendor
library (MIOpen) — run on torch's CPU convolutions with the same weights.

  mvsnet_forward    MVSNet.forward     rmvd/models/mvsnet.py:45-168
  robustmvd_forward RobustMVD.forward  rmvd/models/robust_mvd.py:57-99

Inputs are what the models' input_adapter produces (normalised images, pixel/relative intrinsics, poses),
as numpy arrays with a leading batch axis; `sd` is a {key: float32 ndarray} state dict.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import c_oracle as CO
from . import mvd_oracle as O


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _bn2d(x, sd, p, eps=1e-5):
    return F.batch_norm(x, _t(sd[p + "running_mean"]), _t(sd[p + "running_var"]), _t(sd[p + "weight"]), _t(sd[p + "bias"]),
                        False, 0.0, eps)


def conv_bn_relu_2d(x, weight, bn=None, bias=None, stride=1, relu=True, eps=1e-5):
    """One ConvBnReLU (mvsnet_components.py:8-22) / plain Conv2d on torch CPU, the reference's own arithmetic: conv with
    padding k/2, eval-mode batch_norm from bn = (weight, bias, running_mean, running_var), ReLU.  x (B,Cin,H,W) numpy."""
    with torch.no_grad():
        w = _t(weight)
        y = F.conv2d(_t(x), w, None if bias is None else _t(bias), stride, w.shape[-1] // 2)
        if bn is not None:
            y = F.batch_norm(y, _t(bn[2]), _t(bn[3]), _t(bn[0]), _t(bn[1]), False, 0.0, eps)
        return (F.relu(y) if relu else y).numpy()


def feature_net(x, sd, prefix="feature."):
    """FeatureNet (mvsnet_components.py:44-66) on torch CPU."""
    spec = [(1, 1), (1, 1), (2, 2), (1, 1), (1, 1), (2, 2), (1, 1)]
    with torch.no_grad():
        x = _t(x)
        for i, (stride, pad) in enumerate(spec):
            w = _t(sd[f"{prefix}conv{i}.conv.weight"])
            x = F.relu(_bn2d(F.conv2d(x, w, None, stride, w.shape[-1] // 2), sd, f"{prefix}conv{i}.bn."))
        return F.conv2d(x, _t(sd[prefix + "feature.weight"]), _t(sd[prefix + "feature.bias"]), 1, 1).numpy()


def mvsnet_forward(images, poses, intrinsics, keyview_idx, depth_range, sd, num_sampling_steps, timings=None):
    """images[v] (B,3,H,W) normalised; poses[v] (B,4,4); intrinsics[v] (B,3,3); depth_range (min, max) scalars."""
    import time
    B = images[0].shape[0]
    D = num_sampling_steps
    depth = np.stack([np.linspace(np.float32(depth_range[0]), np.float32(depth_range[1]), D, dtype=np.float32)] * B)
    order = [keyview_idx] + [v for v in range(len(images)) if v != keyview_idx]
    t0 = time.perf_counter()
    feats = [feature_net(images[v], sd) for v in order]
    t1 = time.perf_counter()
    projs = []
    for b in range(B):
        projs.append(O.mvsnet_proj_matrices([intrinsics[v][b] for v in range(len(images))],
                                            [poses[v][b] for v in range(len(images))], keyview_idx))
    P = [np.stack([projs[b][v] for b in range(B)]) for v in order]
    var = CO.warp_variance(feats[0], feats[1:], P[1:], P[0], depth)
    t2 = time.perf_counter()
    cost = CO.cost_reg_net(var, sd, "cost_regularization.")[:, 0]
    t3 = time.perf_counter()
    dep, conf = CO.softmax_regress(cost, depth)
    t4 = time.perf_counter()
    if timings is not None:
        timings.update(features=t1 - t0, warp_variance=t2 - t1, cost_reg=t3 - t2, regress=t4 - t3)
    return {"depth": dep[:, None], "depth_uncertainty": (1 - conf)[:, None]}


def _conv(x, sd, name, stride=1, act=True):
    w = _t(sd[name + ".0.weight"])
    y = F.conv2d(x, w, _t(sd[name + ".0.bias"]), stride, (w.shape[-1] - 1) // 2)
    return F.leaky_relu(y, 0.2) if act else y


def robustmvd_forward(images, poses, intrinsics, keyview_idx, sd, timings=None):
    """images[v] (N,3,H,W) already `/255 - 0.4`; intrinsics relative; returns pred dict like RobustMVD.forward."""
    import time
    order = [v for v in range(len(images)) if v != keyview_idx]
    with torch.no_grad():
        def encoder(img):
            c1 = _conv(_t(img), sd, "encoder.conv1", 2)
            c2 = _conv(c1, sd, "encoder.conv2", 2)
            return c1, c2, _conv(c2, sd, "encoder.conv3", 2)

        t0 = time.perf_counter()
        c1, c2, enc_key = encoder(images[keyview_idx])
        enc_src = [encoder(images[v])[2] for v in order]
        ctx = _conv(enc_key, sd, "context_encoder.conv_redir")
        t1 = time.perf_counter()
        inv = O.compute_sampling_invdepths(0.4, 1000.0, 256)
        corrs, masks = CO.sweep_corr(enc_key.numpy(), [e.numpy() for e in enc_src], intrinsics[keyview_idx],
                                     [intrinsics[v] for v in order], [poses[v] for v in order], inv)
        t2 = time.perf_counter()
        if len(corrs) > 1:
            scores = []
            for c in corrs:
                hid = F.relu(F.conv2d(_t(c), _t(sd["fusion_block.corr_to_view_weight.0.weight"]),
                                      _t(sd["fusion_block.corr_to_view_weight.0.bias"]), 1, 1))
                scores.append(F.conv2d(hid, _t(sd["fusion_block.corr_to_view_weight.2.weight"]),
                                       _t(sd["fusion_block.corr_to_view_weight.2.bias"])).numpy())
            fused, _ = CO.fuse_views(corrs, masks, scores)
        else:
            fused = corrs[0]
        t3 = time.perf_counter()
        enc = {"conv1": c1, "conv2": c2}
        x = torch.cat([ctx, _t(fused)], 1)
        for name, stride in (("conv3_1", 1), ("conv4", 2), ("conv4_1", 1), ("conv5", 2), ("conv5_1", 1), ("conv6", 2),
                             ("conv6_1", 1)):
            x = _conv(x, sd, "fusion_enc_block." + name, stride)
            enc[name] = x

        def head(t, name):
            p = _conv(t, sd, "decoder." + name, act=False)
            return torch.cat([F.relu(p[:, :1]), torch.sigmoid(p[:, 1:] * 0.2) * 20 - 10], 1)

        feat, pred = x, head(x, "pred_0")
        for lvl, skip in enumerate(["conv5_1", "conv4_1", "conv3_1", "conv2", "conv1"], start=1):
            up = F.leaky_relu(F.conv_transpose2d(feat, _t(sd[f"decoder.deconv_{lvl}.0.weight"]),
                                                 _t(sd[f"decoder.deconv_{lvl}.0.bias"]), 2, 1), 0.2)
            pup = F.interpolate(pred, size=up.shape[-2:], mode="bilinear", align_corners=False)
            feat = _conv(torch.cat((enc[skip], up, pup), 1), sd, f"decoder.rfeat{lvl}")
            pred = head(feat, f"pred_{lvl}")
        t4 = time.perf_counter()
        invdepth, log_b = pred[:, 0:1].numpy(), pred[:, 1:2].numpy()
    if timings is not None:
        timings.update(encoders=t1 - t0, sweep_corr=t2 - t1, fusion=t3 - t2, decoder=t4 - t3)
    return {"depth": 1 / (invdepth + 1e-9), "depth_uncertainty": np.exp(log_b) / (invdepth + 1e-9),
            "invdepth": invdepth, "invdepth_log_b": log_b}
