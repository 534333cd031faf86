"""The two models whose forward pass contains the plane-sweep path, with the reference's protocol
(input_adapter / forward / output_adapter, registry entry points) and state-dict keys:

  RobustMVD   rmvd/models/robust_mvd.py:26-158   Path A: K1 sweep-correlation + K2 learned fusion inside a DispNet
  MVSNet      rmvd/models/mvsnet.py:31-217       Path B: K3 warp+variance, K4 CostRegNet, K5 soft argmin
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .blocks import (CostRegNet, DispnetContextEncoder, DispnetCostvolumeEncoder, DispnetDecoder, DispnetEncoder,
                     FeatureNet, LearnedFusion, PlanesweepCorrelation)
from .registry import build_model_with_cfg, register_model
from .utils import exclude_index, get_torch_model_device, select_by_index, to_numpy, to_torch


def _require_multiple(images, m, what):
    h, w = images[0].shape[-2:]
    if h % m or w % m:
        raise NotImplementedError(
            f"{what}: input {h}x{w} is not a multiple of {m}. The reference resizes with skimage.transform.resize "
            "(rmvd/data/transforms.py:56-74), which is not available here; resize the images (and scale the "
            "intrinsics) before calling the model.")


class RobustMVD(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = DispnetEncoder()
        self.context_encoder = DispnetContextEncoder()
        self.corr_block = PlanesweepCorrelation()
        self.fusion_block = LearnedFusion()
        self.fusion_enc_block = DispnetCostvolumeEncoder()
        self.decoder = DispnetDecoder()
        for m in self.modules():  # robust_mvd.py:39-55
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.kaiming_normal_(m.weight, a=0.2, nonlinearity="leaky_relu")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, images, poses, intrinsics, keyview_idx, **_):
        if isinstance(keyview_idx, torch.Tensor):
            keyview_idx = keyview_idx.tolist() if keyview_idx.dim() else int(keyview_idx)
        image_key = select_by_index(images, keyview_idx)
        images_source = exclude_index(images, keyview_idx)
        intrinsics_key = select_by_index(intrinsics, keyview_idx)
        intrinsics_source = exclude_index(intrinsics, keyview_idx)
        source_to_key = exclude_index(poses, keyview_idx)

        n = image_key.shape[0]
        same = all(im.shape == image_key.shape for im in images_source)
        if same:  # one encoder pass over key + all sources
            feats = self.encoder.conv3(self.encoder.conv2(self.encoder.conv1(torch.cat(images_source, 0))))
            enc_sources = list(torch.split(feats, n, 0))
        else:
            enc_sources = [self.encoder(im)[1] for im in images_source]
        all_enc_key, enc_key = self.encoder(image_key)
        ctx = self.context_encoder(enc_key)

        corrs, masks, _ = self.corr_block(
            feat_key=enc_key, intrinsics_key=intrinsics_key, feat_sources=enc_sources,
            source_to_key_transforms=source_to_key, intrinsics_sources=intrinsics_source,
            num_sampling_points=256, min_depth=0.4, max_depth=1000.0)  # robust_mvd.py:77-79
        fused_corr, _ = self.fusion_block(corrs=corrs, masks=masks)
        all_enc_fused, enc_fused = self.fusion_enc_block(corr=fused_corr, ctx=ctx)
        dec = self.decoder(enc_fused=enc_fused, all_enc={**all_enc_key, **all_enc_fused})

        pred = {"depth": 1 / (dec["invdepth"] + 1e-9),
                "depth_uncertainty": torch.exp(dec["invdepth_log_b"]) / (dec["invdepth"] + 1e-9)}
        aux = dec
        aux["depth"], aux["depth_uncertainty"] = pred["depth"], pred["depth_uncertainty"]
        return pred, aux

    def input_adapter(self, images, keyview_idx, poses, intrinsics, **_):
        device = get_torch_model_device(self)
        _require_multiple(images, 64, "robust_mvd")
        ht, wd = images[0].shape[-2:]
        images = [im / 255.0 - 0.4 for im in images]
        scale = np.array([[wd] * 3, [ht] * 3, [1.0] * 3], dtype=np.float32)  # relative intrinsics, :118-120
        intrinsics = [k / scale for k in intrinsics]
        images, keyview_idx, poses, intrinsics = to_torch((images, keyview_idx, poses, intrinsics), device=device)
        images = [im.float() for im in images]
        poses = [p.float() for p in poses]
        intrinsics = [k.float() for k in intrinsics]
        return {"images": images, "keyview_idx": keyview_idx, "poses": poses, "intrinsics": intrinsics}

    def output_adapter(self, model_output):
        pred, aux = model_output
        return to_numpy(pred), to_numpy(aux)


class MVSNet(nn.Module):
    def __init__(self, sample_in_inv_depth_space=False, num_sampling_steps=192):
        super().__init__()
        if sample_in_inv_depth_space:
            raise NotImplementedError("sample_in_inv_depth_space=True is a dead branch in the reference "
                                      "(tensor[::-1] raises, mvsnet.py:50,56-63)")
        self.feature = FeatureNet()
        self.cost_regularization = CostRegNet()
        self.num_sampling_steps = num_sampling_steps
        self.sample_in_inv_depth_space = False

    def depth_samples(self, depth_range, n, device):
        lo, hi = (0.2, 100.0) if depth_range is None else (float(depth_range[0][0]), float(depth_range[1][0]))
        d = torch.linspace(lo, hi, self.num_sampling_steps, dtype=torch.float32)  # batch element 0's range, :66-73
        return torch.stack(n * [d]).to(device)

    @staticmethod
    def projection_matrices(intrinsics, poses, keyview_idx):
        """mvsnet.py:76-103: K[:2] *= 0.25; P[:3,:4] = K @ pose[:3,:4]; the key view's P is inverted.
        (The reference writes into the caller's pose tensors; here they are left untouched.)"""
        out = []
        for v, (K, T) in enumerate(zip(intrinsics, poses)):
            K = K.float() * torch.tensor([[0.25] * 3, [0.25] * 3, [1.0] * 3], device=K.device)
            P = T.float().clone()
            P[:, :3, :4] = torch.matmul(K, P[:, :3, :4])
            is_key = torch.as_tensor(keyview_idx, device=P.device).reshape(-1) == v
            if bool(is_key.any()):
                P = torch.where(is_key.view(-1, 1, 1), torch.inverse(P), P)
            out.append(P)
        return out

    def forward(self, images, poses, intrinsics, keyview_idx, depth_range=None, **_):
        if isinstance(keyview_idx, torch.Tensor):
            keyview_idx = keyview_idx.tolist() if keyview_idx.dim() else int(keyview_idx)
        n = images[0].shape[0]
        device = images[0].device
        depth_samples = self.depth_samples(depth_range, n, device)
        proj = self.projection_matrices(intrinsics, poses, keyview_idx)
        views = [select_by_index(images, keyview_idx)] + exclude_index(images, keyview_idx)
        projs = [select_by_index(proj, keyview_idx)] + exclude_index(proj, keyview_idx)

        feats = self.feature(torch.cat(views, 0))           # (V*B, 32, h, w) on MIOpen
        feats = list(torch.split(feats, n, 0))
        var = ops.warp_variance(feats[0], feats[1:], projs[1:], projs[0], depth_samples, channels_last=True)   # K3
        cost = self.cost_regularization.forward_channels_last(var)                                              # K4
        del var
        depth, conf = ops.softmax_regress(cost, depth_samples)                                                  # K5
        pred = {"depth": depth.unsqueeze(1), "depth_uncertainty": (1 - conf).unsqueeze(1)}
        return pred, {}

    def input_adapter(self, images, keyview_idx, poses=None, intrinsics=None, depth_range=None, masks=None):
        device = get_torch_model_device(self)
        _require_multiple(images, 32, "mvsnet")
        # images are 0..255: /255 (NormalizeImagesToMinMax(0,1) is a plain rescale, transforms.py:283-288),
        # then ImageNet shift/scale (mvsnet.py:181-183)
        mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(-1, 1, 1)
        std = np.array([0.229, 0.224, 0.225], np.float32).reshape(-1, 1, 1)
        images = [((im / 255.0 - mean) / std).astype(np.float32) for im in images]
        images, keyview_idx, intrinsics, poses, depth_range, masks = to_torch(
            (images, keyview_idx, intrinsics, poses, depth_range, masks), device=device)
        return {"images": [im.float() for im in images], "poses": poses, "intrinsics": intrinsics,
                "keyview_idx": keyview_idx, "depth_range": depth_range, "masks": masks}

    def output_adapter(self, model_output):
        pred, aux = model_output
        return to_numpy(pred), to_numpy(aux)


@register_model
def robust_mvd(pretrained=True, weights=None, train=False, num_gpus=1, **kwargs):
    if pretrained and weights is None:
        raise RuntimeError("robust_mvd: the pretrained weights are URL-only (robust_mvd.py:153) and there is no network; "
                           "pass weights=<path to robustmvd_600k.pt> or pretrained=False")
    return build_model_with_cfg(model_cls=RobustMVD, weights=weights, train=train, num_gpus=num_gpus)


@register_model(trainable=False)
def robust_mvd_5M(pretrained=True, weights=None, train=False, num_gpus=1, **kwargs):
    if pretrained and weights is None:
        raise RuntimeError("robust_mvd_5M: the pretrained weights are URL-only (robust_mvd.py:142); pass weights=<path>")
    return build_model_with_cfg(model_cls=RobustMVD, weights=weights, train=train, num_gpus=num_gpus)


@register_model(trainable=False)
def mvsnet_train(pretrained=True, weights=None, train=False, num_gpus=1, **kwargs):
    assert not (pretrained and weights is None), "Pretrained weights are not available for this model."
    cfg = {"sample_in_inv_depth_space": False, "num_sampling_steps": 256}
    return build_model_with_cfg(model_cls=MVSNet, cfg=cfg, weights=weights, train=train, num_gpus=num_gpus, **kwargs)
