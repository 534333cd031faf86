"""The two models whose forward pass contains the plane-sweep path, with the reference's protocol
(input_adapter / forward / output_adapter, registry entry points) and state-dict keys:

  RobustMVD   rmvd/models/robust_mvd.py:26-158   Path A: K1 sweep-correlation + K2 learned fusion inside a DispNet
  MVSNet      rmvd/models/mvsnet.py:31-217       Path B: K3 warp+variance, K4 CostRegNet, K5 soft argmin
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .blocks import (CostRegNet, DispnetContextEncoder, DispnetCostvolumeEncoder, DispnetDecoder, DispnetEncoder,
                     FeatureNet, LearnedFusion, PlanesweepCorrelation)
from .registry import build_model_with_cfg, register_model
from .utils import exclude_index, get_torch_model_device, select_by_index, to_numpy, to_torch


def _key_positions(keyview_idx, n):
    """keyview_idx (int, list, numpy or tensor) -> per-sample python ints.  The adapters keep it on the host; a
    GPU tensor (the training harness moves whole samples to the device) costs one synchronising copy."""
    if isinstance(keyview_idx, torch.Tensor):
        keyview_idx = keyview_idx.detach().cpu().numpy()
    k = np.asarray(keyview_idx).reshape(-1)
    if k.size == 1:
        return [int(k[0])] * n
    if k.size != n:
        raise ValueError(f"keyview_idx has {k.size} entries for a batch of {n}")
    return [int(v) for v in k]


def _upscale_to_multiple(images, intrinsics, m, device):
    """UpscaleInputsToNextMultipleOf(m) / the ResizeInputs call of robust_mvd.py:104-113 on the device (SURVEY.md 8f
    rank 2).  images: list of (N,3,H,W) arrays or tensors in 0..255; intrinsics: list of (N,3,3) or None.  Uploads the
    raw images, resizes them with the engine's order-1 kernel (ops.resize_order1: skimage.transform.resize's arithmetic
    for upscaling) when H or W is not a multiple of m, and scales the intrinsics by [[wd/orig_wd]*3, [ht/orig_ht]*3,
    [1]*3] in float32 like transforms.py:69-72.  Returns (device images, intrinsics, ht, wd)."""
    orig_ht, orig_wd = images[0].shape[-2:]
    ht, wd = int(math.ceil(orig_ht / m) * m), int(math.ceil(orig_wd / m) * m)
    images = [im.float() for im in to_torch(list(images), device=device)]
    if ht != orig_ht or wd != orig_wd:
        if any(tuple(im.shape[-2:]) != (orig_ht, orig_wd) for im in images):
            raise ValueError("all views must have the size of the first one to be resized together (transforms.py:56-66)")
        images = [ops.resize_order1(im, ht, wd) for im in images]
        if intrinsics is not None:
            scale = np.array([[wd / orig_wd] * 3, [ht / orig_ht] * 3, [1.0] * 3], dtype=np.float32)
            intrinsics = [(k.detach().cpu().numpy() if isinstance(k, torch.Tensor) else np.asarray(k)) * scale
                          for k in intrinsics]
    return images, intrinsics, ht, wd


def _stack_views(images, fn):
    """fn(image) for every view, written into ONE buffer when the views have equal shapes, and returned as its slices: what the
    forwards batch over (`torch.cat` of the views, robust_mvd.py / mvsnet.py) is then a zero-copy view (`_as_batch`).
    fn's result is converted to float32 by the copy into the buffer (one rounding, as `.float()` would)."""
    if len(images) < 2 or any(im.shape != images[0].shape for im in images):
        return [fn(im).float() for im in images]
    buf = torch.empty((len(images),) + tuple(images[0].shape), dtype=torch.float32, device=images[0].device)
    for i, im in enumerate(images):
        buf[i].copy_(fn(im))
    return [buf[i] for i in range(len(images))]


def _as_batch(views):
    """torch.cat(views, 0) — without the copy when the views are consecutive slices of one buffer (`_stack_views`)."""
    v0 = views[0]
    if len(views) == 1:
        return v0
    step = v0.numel() * v0.element_size()
    if all(v.shape == v0.shape and v.dtype == v0.dtype and v.is_contiguous() and v.untyped_storage().data_ptr() == v0.untyped_storage().data_ptr()
           and v.data_ptr() == v0.data_ptr() + i * step for i, v in enumerate(views)):
        return torch.as_strided(v0, (len(views) * v0.shape[0],) + tuple(v0.shape[1:]), v0.stride())
    return torch.cat(list(views), 0)


class RobustMVD(nn.Module):
    def __init__(self, half_dispnet=False, engine_dispnet=True):
        """engine_dispnet (default on): at inference on the GPU in fp32 the whole 2-D CNN around the sweep runs on the engine's
        split-operand convolution kernels, channel-last from the first layer to the prediction heads (dispnet_engine.py);
        False: the reference's layer-by-layer form on the vendor library's convolutions.
        half_dispnet (an extension; the reference has no such switch; SURVEY.md 8f rank 1 names fp16 as a tuning lever
        of the DispNet row): at inference the 2-D convolutions around the sweep — encoder, context encoder, fusion score
        convs, cost-volume encoder, decoder — run under torch.autocast(float16) on the vendor library's fp16 kernels
        (fp32 accumulation); the sweep (K1), the fusion arithmetic (K2) and the prediction heads' final arithmetic stay
        fp32.  Default False = fp32 everywhere, like the reference."""
        super().__init__()
        self.half_dispnet = bool(half_dispnet)
        self.engine_dispnet = bool(engine_dispnet)
        self._engine = None
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

    SWEEP = dict(num_sampling_points=256, min_depth=0.4, max_depth=1000.0)  # robust_mvd.py:77-79

    def _prepare(self):
        """Constants of the sweep uploaded ahead of the first forward (FramePipeline calls this before it puts frames on
        side streams)."""
        dev = next(self.parameters()).device
        if dev.type == "cuda":
            self.corr_block.warm(sampling_type="linear_invdepth", device=dev, **self.SWEEP)

    def forward(self, images, poses, intrinsics, keyview_idx, **_):
        half = self.half_dispnet and images[0].is_cuda and not torch.is_grad_enabled()
        with torch.autocast("cuda", dtype=torch.float16, enabled=half):
            pred, aux = self._forward(images, poses, intrinsics, keyview_idx)
        if half:  # the public outputs are float32 like the reference's
            f32 = lambda v: [x.float() for x in v] if isinstance(v, list) else v.float()
            pred = {k: f32(v) for k, v in pred.items()}
            aux = {k: f32(v) for k, v in aux.items()}
        return pred, aux

    def _forward(self, images, poses, intrinsics, keyview_idx):
        key_pos = _key_positions(keyview_idx, images[0].shape[0])
        keyview_idx = key_pos[0] if all(k == key_pos[0] for k in key_pos) else key_pos
        image_key = select_by_index(images, keyview_idx)
        images_source = exclude_index(images, keyview_idx)
        intrinsics_key = select_by_index(intrinsics, keyview_idx)
        intrinsics_source = exclude_index(intrinsics, keyview_idx)
        source_to_key = exclude_index(poses, keyview_idx)

        n = image_key.shape[0]
        same = all(im.shape == image_key.shape for im in images_source)
        if (self.engine_dispnet and same and image_key.is_cuda and image_key.dtype == torch.float32 and not torch.is_grad_enabled()
                and not torch.is_autocast_enabled() and image_key.shape[-2] % 64 == 0 and image_key.shape[-1] % 64 == 0):
            if self._engine is None:
                from .dispnet_engine import DispnetEngine
                self._engine = DispnetEngine(self)
            dec = self._engine.forward(image_key, images_source, intrinsics_key, intrinsics_source, source_to_key)
            return self._outputs(dec)
        if same:  # one encoder pass over key + all sources
            feats = self.encoder.conv3(self.encoder.conv2(self.encoder.conv1(_as_batch(images_source))))
            enc_sources = list(torch.split(feats, n, 0))
        else:
            enc_sources = [self.encoder(im)[1] for im in images_source]
        all_enc_key, enc_key = self.encoder(image_key)
        ctx = self.context_encoder(enc_key)

        corrs, masks, _ = self.corr_block(
            feat_key=enc_key, intrinsics_key=intrinsics_key, feat_sources=enc_sources,
            source_to_key_transforms=source_to_key, intrinsics_sources=intrinsics_source,
            **self.SWEEP)
        fused_corr, _ = self.fusion_block(corrs=corrs, masks=masks)
        all_enc_fused, enc_fused = self.fusion_enc_block(corr=fused_corr, ctx=ctx)
        dec = self.decoder(enc_fused=enc_fused, all_enc={**all_enc_key, **all_enc_fused})
        return self._outputs(dec)

    @staticmethod
    def _outputs(dec):
        inv, log_b = dec["invdepth"].float(), dec["invdepth_log_b"].float()
        pred = {"depth": 1 / (inv + 1e-9), "depth_uncertainty": torch.exp(log_b) / (inv + 1e-9)}
        aux = dec
        aux["depth"], aux["depth_uncertainty"] = pred["depth"], pred["depth_uncertainty"]
        return pred, aux

    def input_adapter(self, images, keyview_idx, poses, intrinsics, **_):
        device = get_torch_model_device(self)
        # resize to the next multiple of 64 if needed (robust_mvd.py:104-113), on the device
        images, intrinsics, ht, wd = _upscale_to_multiple(images, intrinsics, 64, device)
        scale = np.array([[wd] * 3, [ht] * 3, [1.0] * 3], dtype=np.float32)  # relative intrinsics, :118-120
        intrinsics = [k / scale for k in intrinsics]
        poses, intrinsics = to_torch((poses, intrinsics), device=device)
        keyview_idx = to_torch(keyview_idx)  # stays on the host: only used to order the views
        # im / 255 - 0.4 (robust_mvd.py:113-116) on the device: the raw images are uploaded, the arithmetic is the
        # reference's float32 operations one by one (true division by a tensor, not torch's scalar-reciprocal shortcut)
        c255 = torch.full((1,), 255.0, dtype=torch.float32, device=device)
        images = _stack_views(images, lambda im: im / c255 - 0.4)
        poses = [p.float() for p in poses]
        intrinsics = [k.float() for k in intrinsics]
        return {"images": images, "keyview_idx": keyview_idx, "poses": poses, "intrinsics": intrinsics}

    def output_adapter(self, model_output):
        pred, aux = model_output
        return to_numpy(pred), to_numpy(aux)


class MVSNet(nn.Module):
    def __init__(self, sample_in_inv_depth_space=False, num_sampling_steps=192, half_features=False, conv0_split=True,
                 exact_grid=False):
        """half_features (an extension; the reference has no such switch): BASELINE.json configs[3] — the feature maps
        are rounded to fp16 before the sweep, the variance volume is stored fp16 and the regulariser's first layer runs
        on fp16 MFMA with fp32 accumulation; everything else (positions, blend, variance, layers 2..11, soft argmin)
        stays fp32.  Regressed depth within rtol 1e-2 of the fp32 path.  Plain IEEE fp16 without scaling:
        feature or variance magnitudes beyond 65504 become inf (include/mvd.h); meant for O(1) features.
        conv0_split (default True): the regulariser's first layer with split, range-scaled fp16 operands on fp16 MFMA
        (CostRegNet); False = fp32 MFMA.  exact_grid: K3's sampling positions by the reference's own rounding chain."""
        super().__init__()
        self.half_features = bool(half_features)
        if sample_in_inv_depth_space:
            raise NotImplementedError("sample_in_inv_depth_space=True is a dead branch in the reference "
                                      "(tensor[::-1] raises, mvsnet.py:50,56-63)")
        self.feature = FeatureNet(split_layers=conv0_split)  # same switch: False = every layer on the fp32 matrix instruction
        self.cost_regularization = CostRegNet(conv0_split=conv0_split)  # split-operand first layer, see CostRegNet
        self.exact_grid = bool(exact_grid)  # K3's sampling positions by the reference's own rounding chain (blocks/utils.py:234-266)
        self.num_sampling_steps = num_sampling_steps
        self.sample_in_inv_depth_space = False
        self._intrinsics_scale_host = torch.tensor([[0.25] * 3, [0.25] * 3, [1.0] * 3])  # feature maps are 1/4 resolution
        self.register_buffer("intrinsics_scale", self._intrinsics_scale_host.clone(), persistent=False)

    def depth_samples(self, depth_range, n, device):
        """linspace(min[0], max[0], D) of batch element 0's range (mvsnet.py:66-73), computed on the device
        with torch.linspace's own formula (start + i*step below the midpoint, end - (D-1-i)*step above)
        so that a range that lives on the GPU never has to come back to the host."""
        D = self.num_sampling_steps
        if depth_range is None:
            return torch.linspace(0.2, 100.0, D, dtype=torch.float32, device=device).expand(n, D).contiguous()
        lo, hi = depth_range[0], depth_range[1]
        if not (isinstance(lo, torch.Tensor) and lo.is_cuda) and not (isinstance(hi, torch.Tensor) and hi.is_cuda):
            # host-side range (what this package's input_adapter hands over): torch.linspace itself, bit for bit
            lo = float(torch.as_tensor(lo, dtype=torch.float32).reshape(-1)[0])
            hi = float(torch.as_tensor(hi, dtype=torch.float32).reshape(-1)[0])
            d = torch.linspace(lo, hi, D, dtype=torch.float32)
            return d.expand(n, D).contiguous().to(device, non_blocking=True)
        lo = torch.as_tensor(lo, dtype=torch.float32, device=device).reshape(-1)[0]
        hi = torch.as_tensor(hi, dtype=torch.float32, device=device).reshape(-1)[0]
        i = torch.arange(D, dtype=torch.float32, device=device)
        step = (hi - lo) / (D - 1)
        d = torch.where(i < D // 2, lo + step * i, hi - step * (D - 1 - i))
        return d.expand(n, D).contiguous()

    def projection_matrices(self, intrinsics, poses, key_pos, device):
        """mvsnet.py:76-103: K[:2] *= 0.25; P[:3,:4] = K @ pose[:3,:4]; the key view's P is inverted.
        key_pos: per-sample python ints.  Calibration that is still on the host (what this package's
        input_adapter hands over) is processed there — a handful of 4x4 products — and uploaded as ONE packed
        tensor; calibration that already lives on the GPU is processed on the GPU without host synchronisation
        (inv_ex does not check `info`).  The reference writes into the caller's pose tensors; here they are untouched."""
        on_host = not intrinsics[0].is_cuda and not poses[0].is_cuda
        scale = self._intrinsics_scale_host if on_host else self.intrinsics_scale
        out = []
        for v, (K, T) in enumerate(zip(intrinsics, poses)):
            P = T.float().clone()
            P[:, :3, :4] = torch.matmul(K.float() * scale, P[:, :3, :4])
            if any(k == v for k in key_pos):
                inv = torch.linalg.inv(P) if on_host else torch.linalg.inv_ex(P, check_errors=False).inverse
                if all(k == v for k in key_pos):
                    P = inv
                else:
                    sel = torch.tensor([k == v for k in key_pos], device=P.device).view(-1, 1, 1)
                    P = torch.where(sel, inv, P)
            out.append(P)
        if on_host:
            packed = torch.stack(out, 0).to(device, non_blocking=True)  # (V+1, B, 4, 4)
            out = list(packed.unbind(0))
        return out

    def forward(self, images, poses, intrinsics, keyview_idx, depth_range=None, **_):
        n = images[0].shape[0]
        device = images[0].device
        key_pos = _key_positions(keyview_idx, n)
        kidx = key_pos[0] if all(k == key_pos[0] for k in key_pos) else key_pos
        depth_samples = self.depth_samples(depth_range, n, device)
        proj = self.projection_matrices(intrinsics, poses, key_pos, device)
        views = [select_by_index(images, kidx)] + exclude_index(images, kidx)
        projs = [select_by_index(proj, kidx)] + exclude_index(proj, kidx)

        # K6 x 8: ((V+1)*B, h+3, w+3, 32), the last layer writing straight into K3's zero-bordered staging layout
        split = self.cost_regularization.conv0_split and not self.half_features
        feats = self.feature.forward_layout(_as_batch(views), L.LAYOUT_NHWC_BORDER, return_absmax=split)
        if split:
            feats, a_feat = feats
        if self.half_features:
            feats = ops.to_f16(feats)  # one rounding to fp16 (zero border stays zero)
        feats = list(torch.split(feats, n, 0))
        amax = None
        if self.half_features:
            var = ops.warp_variance_f16(feats[0], feats[1:], projs[1:], projs[0], depth_samples)                          # K3 (fp16)
        else:
            var = ops.warp_variance(feats[0], feats[1:], projs[1:], projs[0], depth_samples, channels_last=True, staged=True,
                                    exact_grid=self.exact_grid)                                                  # K3
            if split:
                # the range the split first layer scales by: a BOUND instead of K3's max |var| by-product (25 us in the frame).  A warped
                # feature is a convex combination of features, and a variance is at most the mean square: var <= (max |feature|)^2;
                # the bound is one or two orders loose, which moves conv0's absolute error floor from 2^-50 to about 2^-44 of max |var|
                amax = a_feat * a_feat
        cost = self.cost_regularization.forward_channels_last(var, x_absmax=amax)                               # K4
        del var
        depth, conf = ops.softmax_regress(cost, depth_samples)                                                  # K5
        pred = {"depth": depth.unsqueeze(1), "depth_uncertainty": (1 - conf).unsqueeze(1)}
        return pred, {}

    def input_adapter(self, images, keyview_idx, poses=None, intrinsics=None, depth_range=None, masks=None):
        device = get_torch_model_device(self)
        # UpscaleInputsToNextMultipleOf(32) (mvsnet.py:178), on the device
        images, intrinsics, _, _ = _upscale_to_multiple(images, intrinsics, 32, device)
        # images are 0..255: /255 (NormalizeImagesToMinMax(0,1) is a plain rescale, transforms.py:283-288),
        # then ImageNet shift/scale (mvsnet.py:181-183)
        masks = to_torch(masks, device=device)
        # The reference's chain (mvsnet.py:181-183 -> transforms.py:283-311) on the device, rounding for rounding:
        # x = im / 255.0 in float32 (a true division, not torch's scalar-reciprocal shortcut), then
        # (x - shift) / scale with shift and scale Python LISTS, i.e. float64 arithmetic, then astype(float32).
        mean = torch.tensor([0.485, 0.456, 0.406], dtype=torch.float64, device=device).view(-1, 1, 1)
        std = torch.tensor([0.229, 0.224, 0.225], dtype=torch.float64, device=device).view(-1, 1, 1)
        c255 = torch.full((1,), 255.0, dtype=torch.float32, device=device)
        images = _stack_views(images, lambda im: ((im.float() / c255).double() - mean) / std)
        # stay on the host: keyview_idx only orders the views, depth_range only seeds torch.linspace, and the 4x4
        # calibration products are cheaper there than as a dozen tiny launches (forward accepts either placement)
        keyview_idx, depth_range, intrinsics, poses = to_torch((keyview_idx, depth_range, intrinsics, poses))
        return {"images": images, "poses": poses, "intrinsics": intrinsics,
                "keyview_idx": keyview_idx, "depth_range": depth_range, "masks": masks}

    def output_adapter(self, model_output):
        pred, aux = model_output
        return to_numpy(pred), to_numpy(aux)


@register_model  # trainable like the reference's: the sweep (K1) and the fusion (K2) back-propagate through the engine's VJP kernels
def robust_mvd(pretrained=True, weights=None, train=False, num_gpus=1, **kwargs):
    if pretrained and weights is None:
        raise RuntimeError("robust_mvd: the pretrained weights are URL-only (robust_mvd.py:153) and there is no network; "
                           "pass weights=<path to robustmvd_600k.pt> or pretrained=False")
    # kwargs: half_dispnet=True selects the fp16-convolution variant (extension; create_model("robust_mvd", half_dispnet=True))
    return build_model_with_cfg(model_cls=RobustMVD, weights=weights, train=train, num_gpus=num_gpus, **kwargs)


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
