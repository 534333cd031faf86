"""Operator-level modules of the plane-sweep path with the reference's names, signatures and
state-dict keys; their forward passes run on the HIP engine (ops.py).

  PlanesweepCorrelation   rmvd/models/blocks/planesweep_corr.py:371-521   -> K1 mvd_sweep_corr_f32
  LearnedFusion           rmvd/models/blocks/learned_fusion.py:6-54       -> MIOpen score convs + K2 mvd_fuse_views_f32
  homo_warp               rmvd/models/blocks/utils.py:222-268             -> mvd_homo_warp_f32
  CostRegNet              rmvd/models/blocks/mvsnet_components.py:69-123  -> K4 mvd_conv3d_bn_relu_f32 x 11
  depth_regression        rmvd/models/blocks/utils.py:271-274             (plain expectation; the fused K5 is ops.softmax_regress)

  FeatureNet              rmvd/models/blocks/mvsnet_components.py:44-66   -> K6 mvd_conv2d_bn_relu_f32 x 8

The DispNet encoder/decoder 2-D CNN around the Path-A sweep is an ordinary torch module that runs on MIOpen.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import ops
from .utils import to_torch

homo_warp = ops.homo_warp


def compute_sampling_invdepths(min_depth, max_depth, num_samples, sampling_type="linear_invdepth"):
    """planesweep_corr.py:524-555. Scalars or per-sample arrays -> (1 or N, num_samples), far to near."""
    def col(v):
        if isinstance(v, (float, int, np.floating)):
            v = torch.tensor([float(v)])
        return to_torch(v).float()[..., None]

    lo, hi = col(min_depth), col(max_depth)
    steps = torch.arange(0, num_samples, dtype=lo.dtype, device=lo.device)[None]
    if sampling_type == "linear_invdepth":
        return 1 / hi + steps * (1 / lo - 1 / hi) / (num_samples - 1)
    if sampling_type == "linear_depth":
        return (1 / (lo + steps * (hi - lo) / (num_samples - 1))).flip(1)
    raise ValueError(f"sampling_type {sampling_type!r}")


class PlanesweepCorrelation(nn.Module):
    """Same call signature and return value as the reference block; no parameters.
    Stateless across calls (the reference keeps per-call state on self and is not re-entrant)."""

    def __init__(self, warp_only=False, normalize="dim"):
        """normalize (TorchCorr, planesweep_corr.py:142-189): "dim" divides the dot products by sqrt(C) (what robust_mvd
        uses); True / "before" L2-normalises both feature maps along C first (x / (|x| + 1e-9), :8-10); False leaves
        the raw dot products.  warp_only=True (WarpOnlyCorr, :106-139) returns the warped source FEATURES (N,S,C,h,w) and
        the sampling mask instead of correlations; there "before" normalises the source features first, True / "after" the
        warped ones along C, "dim" / False nothing (:128-136).  Inference only."""
        super().__init__()
        self.warp_only = bool(warp_only)
        allowed = ("dim", "before", "after", True, False) if warp_only else ("dim", "before", True, False)
        if normalize not in allowed:
            raise ValueError(f"normalize={normalize!r}: expected one of {allowed}")
        self.normalize = normalize
        self._invdepth_cache = {}  # (num, min, max, type, device) -> device tensor: constants of the model, uploaded once

    def warm(self, num_sampling_points, min_depth, max_depth, sampling_type, device):
        """Uploads the sampling inverse depths of a fixed (scalar) range once, ahead of the first forward."""
        ckey = (num_sampling_points, float(min_depth), float(max_depth), sampling_type, str(device))
        if ckey not in self._invdepth_cache:
            self._invdepth_cache[ckey] = compute_sampling_invdepths(min_depth, max_depth, num_sampling_points,
                                                                    sampling_type).to(device)
        return self._invdepth_cache[ckey]

    def forward(self, feat_key, intrinsics_key, feat_sources, source_to_key_transforms, intrinsics_sources=None,
                num_sampling_points=None, min_depth=None, max_depth=None, sampling_invdepths=None,
                sampling_type="linear_invdepth"):
        """Differentiable w.r.t. feat_key and feat_sources like the reference's correlate() (planesweep_corr.py:514-521):
        when autograd is recording and a feature map requires grad, the sweep goes through ops.sweep_corr_autograd (the
        engine's VJP kernel, mvd_sweep_corr_backward_f32); otherwise through the plain inference entry point.  The
        sampling grids and masks are constants in both (the reference computes them under no_grad, :436,464,489)."""
        args = (feat_key, intrinsics_key, feat_sources, source_to_key_transforms, intrinsics_sources, num_sampling_points,
                min_depth, max_depth, sampling_invdepths, sampling_type)
        if self.warp_only:
            if ops.needs_grad(feat_sources):
                raise ValueError("PlanesweepCorrelation(warp_only=True) has no backward in this engine: detach the features")
            with torch.no_grad():
                return self._forward(None, *args)
        if ops.needs_grad(feat_key, feat_sources):
            return self._forward(ops.sweep_corr_autograd, *args)
        with torch.no_grad():
            return self._forward(ops.sweep_corr, *args)

    def _forward(self, sweep, feat_key, intrinsics_key, feat_sources, source_to_key_transforms, intrinsics_sources,
                 num_sampling_points, min_depth, max_depth, sampling_invdepths, sampling_type):
        if intrinsics_sources is None:
            intrinsics_sources = [intrinsics_key] * len(feat_sources)
        if not (len(feat_sources) == len(source_to_key_transforms) == len(intrinsics_sources)):
            raise ValueError("feat_sources, source_to_key_transforms and intrinsics_sources differ in length")
        if min_depth is not None and max_depth is not None:
            if sampling_invdepths is not None or num_sampling_points is None or sampling_type is None:
                raise ValueError("give either (num_sampling_points, min_depth, max_depth) or sampling_invdepths")
            scalars = all(isinstance(v, (float, int, np.floating)) for v in (min_depth, max_depth))
            ckey = (num_sampling_points, float(min_depth), float(max_depth), sampling_type, str(feat_key.device)) if scalars else None
            if ckey is not None and ckey in self._invdepth_cache:
                sampling_invdepths = self._invdepth_cache[ckey]  # no host-to-device copy: forward can be graph-captured
            else:
                sampling_invdepths = compute_sampling_invdepths(min_depth, max_depth, num_sampling_points, sampling_type)
                if ckey is not None:
                    sampling_invdepths = self._invdepth_cache[ckey] = sampling_invdepths.to(feat_key.device)
        elif num_sampling_points is not None or min_depth is not None or max_depth is not None or sampling_invdepths is None:
            raise ValueError("give either (num_sampling_points, min_depth, max_depth) or sampling_invdepths")
        inv = sampling_invdepths.to(feat_key.device)
        if inv.dim() < 2 or inv.dim() > 4:
            raise ValueError("sampling_invdepths must be (N, S), (N, S, H) or (N, S, H, W)")
        while inv.dim() < 4:  # planesweep_corr.py:484-485
            inv = inv.unsqueeze(-1)
        inv_out = inv
        if inv.shape[2] == 1 and inv.shape[3] == 1:
            inv = inv.reshape(inv.shape[0], inv.shape[1])
        else:  # per key pixel: (N,S,H,1) or (N,S,H,W), batch broadcast like the reference's arithmetic would
            inv = inv.expand(feat_key.shape[0], inv.shape[1], feat_key.shape[2], feat_key.shape[3]).contiguous()
        if self.warp_only:
            srcs = list(feat_sources)
            if self.normalize == "before":
                srcs = [f / (torch.linalg.norm(f, dim=1, keepdim=True) + 1e-9) for f in srcs]
            after = self.normalize is True or self.normalize == "after"
            warped, masks = [None] * len(srcs), [None] * len(srcs)
            groups = {}
            for i, f in enumerate(srcs):
                groups.setdefault(tuple(f.shape[-2:]), []).append(i)
            for idxs in groups.values():
                wv, mv = ops.sweep_warp([srcs[i] for i in idxs], intrinsics_key, [intrinsics_sources[i] for i in idxs],
                                        [source_to_key_transforms[i] for i in idxs], inv, feat_key.shape[-2:], after)
                for i, wi, mi in zip(idxs, wv, mv):
                    warped[i], masks[i] = wi, mi
            return warped, masks, inv_out
        corr_scale = None  # 1/sqrt(C)
        if self.normalize != "dim":
            corr_scale = 1.0
            if self.normalize:  # True / "before": x / (|x|_2 + 1e-9) along the channels (planesweep_corr.py:8-10,165-167)
                nrm = lambda x: x / (torch.linalg.norm(x, dim=1, keepdim=True) + 1e-9)
                feat_key, feat_sources = nrm(feat_key), [nrm(f) for f in feat_sources]
        # sources may differ in size: one launch per group of equal (hs, ws)
        corrs = [None] * len(feat_sources)
        masks = [None] * len(feat_sources)
        groups = {}
        for i, f in enumerate(feat_sources):
            groups.setdefault(tuple(f.shape[-2:]), []).append(i)
        for idxs in groups.values():
            c, m = sweep(feat_key, [feat_sources[i] for i in idxs], intrinsics_key,
                         [intrinsics_sources[i] for i in idxs], [source_to_key_transforms[i] for i in idxs], inv, corr_scale)
            for i, ci, mi in zip(idxs, c, m):
                corrs[i], masks[i] = ci, mi
        return corrs, masks, inv_out


class LearnedFusion(nn.Module):
    def __init__(self):
        super().__init__()
        self.corr_to_view_weight = nn.Sequential(
            nn.Conv2d(256, 128, kernel_size=3, stride=1, padding=1, bias=True),
            nn.ReLU(inplace=True),
            nn.Conv2d(128, 1, kernel_size=1, stride=1, padding=0, bias=True),
        )

    def forward(self, corrs, masks):
        if len(corrs) == 1:
            return corrs[0], masks[0]
        n = corrs[0].shape[0]
        scores = self.corr_to_view_weight(torch.cat(list(corrs), 0))  # one batched MIOpen call for all views
        scores = list(torch.split(scores, n, 0))
        if ops.needs_grad(corrs, scores):  # training: the engine's VJP kernel (mvd_fuse_views_backward_f32)
            return ops.fuse_views_autograd(corrs, masks, scores)
        return ops.fuse_views(corrs, masks, scores)


def depth_regression(p, depth_values):
    return torch.sum(p * depth_values.view(*depth_values.shape, 1, 1), 1)


# ------------------------------------------------------------------------------------------------
# MVSNet components
# ------------------------------------------------------------------------------------------------
class ConvBnReLU(nn.Module):
    def __init__(self, cin, cout, kernel_size=3, stride=1, pad=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


def fold_bn(bn):
    """Eval-mode BatchNorm as a per-channel scale/shift."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return scale.detach().contiguous(), (bn.bias - bn.running_mean * scale).detach().contiguous()


class FeatureNet(nn.Module):
    """2-D feature pyramid of MVSNet (mvsnet_components.py:44-66).  Parameters live in ordinary nn.Conv2d /
    nn.BatchNorm2d modules (the reference's checkpoints load); forward folds BN (eval mode) and runs 8 fused HIP
    layers (K6, ops.conv2d_bn_relu) on channel-last activations."""

    SPEC = [(3, 8, 3, 1, 1), (8, 8, 3, 1, 1), (8, 16, 5, 2, 2), (16, 16, 3, 1, 1), (16, 16, 3, 1, 1),
            (16, 32, 5, 2, 2), (32, 32, 3, 1, 1)]

    def __init__(self, split_layers=True):
        """split_layers (default on): conv2 .. conv6 and `feature` run on the split-operand implicit-GEMM kernel of the 2-D engine
        (ops.conv2d_split: fp32-grade, 1.2-1.5x the fp32 matrix instruction on these layers, tools/bench_featurenet_engine.py), the
        BN scale folded into their weights; conv0 and conv1 (full resolution, 3 / 8 input channels: traffic-bound) run as ONE fp32
        vector-ALU launch that keeps their intermediate in LDS (ops.conv2d_head).  False: all eight layers on the fp32 matrix instruction."""
        super().__init__()
        self.inplanes = 32
        self.split_layers = bool(split_layers)
        self.fused_head = True  # with split_layers: conv0 and conv1 as one launch (False: two launches of the fp32-MFMA kernel)
        for i, s in enumerate(self.SPEC):
            setattr(self, f"conv{i}", ConvBnReLU(*s))
        self.feature = nn.Conv2d(32, 32, 3, 1, 1)
        self._packed = None
        self._packed_key = None

    def _prepare(self):
        """Packs weights into MFMA fragment order and folds BN; cached until a parameter changes."""
        key = tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))
        if self._packed is not None and self._packed_key == key:
            return self._packed
        if self.training:
            raise RuntimeError("FeatureNet HIP path folds BatchNorm running statistics: call .eval() first")
        pk = []
        for i, (cin, cout, k, stride, _) in enumerate(self.SPEC):
            m = getattr(self, f"conv{i}")
            w, _, _, _ = ops.pack_conv2d_weights(m.conv.weight.detach())
            pk.append((w, cin, cout, k, stride, *fold_bn(m.bn), True))
        w, _, _, _ = ops.pack_conv2d_weights(self.feature.weight.detach())
        pk.append((w, 32, 32, 3, 1, torch.ones(32, device=w.device), self.feature.bias.detach().contiguous(), False))
        if self.split_layers:  # layers 2 .. 7 for the split-operand kernel: BN scale folded into the weights, shift as the bias
            sp = []
            for i in range(2, 8):
                conv = getattr(self, f"conv{i}").conv if i < 7 else self.feature
                _, _, _, _, stride, scale, shift, relu = pk[i]
                sp.append((ops.pack_conv2d_weights_split(conv.weight.detach() * scale.view(-1, 1, 1, 1), shift, stride=stride), relu))
            pk.append(sp)
            # conv0 -> conv1 in one launch (ops.conv2d_head): weights as [ky][kx][cin][cout]
            hw = [getattr(self, f"conv{i}").conv.weight.detach().permute(2, 3, 1, 0).contiguous() for i in range(2)]
            pk.append((hw[0], pk[0][5], pk[0][6], hw[1], pk[1][5], pk[1][6]))
        self._packed, self._packed_key = pk, key
        return pk

    @ops.inference_only
    def forward_layout(self, x, out_layout, return_absmax=False):
        """x (N,3,H,W) normalised images -> features at H/4 x W/4 in `out_layout` (L.LAYOUT_*).  return_absmax: also max |feature|
        over the batch as a one-element device tensor (a by-product of the last layer's store epilogue)."""
        pk = self._prepare()
        if not self.split_layers:
            for i, (w, cin, cout, k, stride, scale, shift, relu) in enumerate(pk):
                x = ops.conv2d_bn_relu(x, w, cin, cout, k, stride, scale, shift, relu=relu,
                                       out_layout=out_layout if i == len(pk) - 1 else L.LAYOUT_NHWC)
            return (x, ops.absmax(x)) if return_absmax else x
        slots = torch.zeros(8, dtype=torch.float32, device=x.device)  # max-|y| slots of the layers, raised by their producers
        if self.fused_head:  # the 8-channel full-resolution intermediate never leaves the chip; max |conv1| from per-tile maxima
            x, a_in = ops.conv2d_head(x, *pk[9], return_absmax=True)
        else:
            for i in range(2):
                w, cin, cout, k, stride, scale, shift, relu = pk[i]
                x = ops.conv2d_bn_relu(x, w, cin, cout, k, stride, scale, shift, relu=relu)
            # max |conv1| by a pass of its own (29 us for 141 MB): as a by-product of conv1's store epilogue (out_absmax=) it cost 65 us
            # in the frame, where the maximum grows across the image and many of the layer's 69,000 waves reach the atomic
            a_in = ops.absmax(x)
        for j, (wts, relu) in enumerate(pk[8]):
            last = j == len(pk[8]) - 1
            if not last:
                x = ops.conv2d_split(x, a_in, wts, act=2 if relu else 0, out_absmax=slots[j:j + 1])
                a_in = slots[j:j + 1]
            elif out_layout == L.LAYOUT_NCHW:
                x = ops.conv2d_split(x, a_in, wts, act=0, planar_out=True)
                if return_absmax:
                    slots[7:8] = ops.absmax(x)
            elif out_layout == L.LAYOUT_NHWC_BORDER:  # K3's zero-bordered staging map: the layer writes the interior
                B, h, w_, _ = x.shape
                buf = torch.zeros((B, h + 3, w_ + 3, wts.cout), dtype=torch.float32, device=x.device)
                ops.conv2d_split(x, a_in, wts, act=0, out=buf[:, 1:h + 1, 1:w_ + 1, :], out_absmax=slots[7:8])
                x = buf
            else:
                x = ops.conv2d_split(x, a_in, wts, act=0, out_absmax=slots[7:8])
        if return_absmax:
            return x, slots[7:8]
        return x

    def forward(self, x):
        """Reference layout: (N,3,H,W) -> (N,32,H/4,W/4)."""
        return self.forward_layout(x, L.LAYOUT_NCHW)


class ConvBnReLU3D(nn.Module):
    def __init__(self, cin, cout, kernel_size=3, stride=1, pad=1):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm3d(cout)


def _deconv_block(cin, cout):
    return nn.Sequential(nn.ConvTranspose3d(cin, cout, kernel_size=3, padding=1, output_padding=1, stride=2, bias=False),
                         nn.BatchNorm3d(cout), nn.ReLU(inplace=True))


class CostRegNet(nn.Module):
    """3-D U-Net regulariser.  Parameters live in ordinary nn.Conv3d / nn.BatchNorm3d modules (so the
    reference's checkpoints load); forward folds BN (eval mode) and runs 11 fused HIP layers on
    channel-last activations."""

    LAYERS = [("conv0", 32, 8, 1), ("conv1", 8, 16, 2), ("conv2", 16, 16, 1), ("conv3", 16, 32, 2), ("conv4", 32, 32, 1),
              ("conv5", 32, 64, 2), ("conv6", 64, 64, 1)]
    UPS = [("conv7", 64, 32), ("conv9", 32, 16), ("conv11", 16, 8)]

    def __init__(self, conv0_split=True):
        """conv0_split (default on): the fp32 operands of the stride-1 layers with 16 or 32 input channels (conv0: 32 -> 8 at
        full resolution, conv2: 16 -> 16, conv4: 32 -> 32) are split into two fp16 terms each, after an exact
        power-of-two range scaling, and multiplied on fp16 MFMA with fp32 accumulation (ops.conv3d_bn_relu_split); its
        results are at least as close to the exact convolution as the fp32 matrix instruction's (measured against a
        float64 oracle over magnitudes 1e-30 .. 1e30: tests/test_hip_f16.py).  False: every layer on fp32 MFMA."""
        super().__init__()
        self.conv0_split = bool(conv0_split)
        for name, cin, cout, stride in self.LAYERS:
            setattr(self, name, ConvBnReLU3D(cin, cout, stride=stride))
        for name, cin, cout in self.UPS:
            setattr(self, name, _deconv_block(cin, cout))
        self.prob = nn.Conv3d(8, 1, 3, stride=1, padding=1)
        self._packed = None
        self._packed_key = None

    def _prepare(self):
        """Packs weights into MFMA fragment order and folds BN; cached until a parameter changes."""
        key = tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))
        if self._packed is not None and self._packed_key == key:
            return self._packed
        if self.training:
            raise RuntimeError("CostRegNet HIP path folds BatchNorm running statistics: call .eval() first")
        pk = {}
        for name, cin, cout, stride in self.LAYERS:
            m = getattr(self, name)
            mode = L.CONV3D_STRIDE1 if stride == 1 else L.CONV3D_STRIDE2
            w, _, _ = ops.pack_conv3d_weights(m.conv.weight.detach(), mode)
            pk[name] = (w, cin, cout, *fold_bn(m.bn), mode)
        for name, cin, cout in self.UPS:
            m = getattr(self, name)
            w, _, _ = ops.pack_conv3d_weights(m[0].weight.detach(), L.DECONV3D_STRIDE2)
            pk[name] = (w, cin, cout, *fold_bn(m[1]), L.DECONV3D_STRIDE2)
        w, _, _ = ops.pack_conv3d_weights(self.prob.weight.detach(), L.CONV3D_STRIDE1)
        pk["prob"] = (w, 8, 1, torch.ones(1, device=w.device), self.prob.bias.detach().contiguous(), L.CONV3D_STRIDE1)
        # fp16-feature variant (BASELINE configs[3]): conv0's weights rounded to fp16 in fp16-MFMA fragment order
        pk["conv0_f16"] = ops.pack_conv3d_weights_f16(self.conv0.conv.weight.detach())
        if self.conv0_split:  # split-operand kernels: conv0 on the plane-marching one, conv1 .. conv6 on the implicit GEMM
            pk["conv0_split"] = ops.pack_conv3d_weights_split(self.conv0.conv.weight.detach())
            sc0, sh0 = pk["conv0"][3], pk["conv0"][4]
            pk["conv0_bound"] = ((self.conv0.conv.weight.detach().abs().sum(dim=(1, 2, 3, 4)) * sc0.abs()).contiguous(), sh0.abs().contiguous())
            for name, _, _, stride in self.LAYERS:
                if name != "conv0":
                    pk[name + "_igemm"] = ops.pack_conv3d_weights_igemm(getattr(self, name).conv.weight.detach(),
                                                                        L.CONV3D_STRIDE1 if stride == 1 else L.CONV3D_STRIDE2)
        self._packed, self._packed_key = pk, key
        return pk

    @ops.inference_only
    def forward_channels_last(self, x, x_absmax=None):
        """x (B,D,h,w,32) fp32, or fp16 (the fp16-feature variant: conv0 then runs on fp16 MFMA) -> cost (B,D,h,w) fp32
        (the single output channel squeezed).  x_absmax: max |x| as a one-element device tensor when the producer knows it
        (ops.warp_variance(..., return_absmax=True)); the split first layer otherwise measures it with one more pass."""
        if x.shape[1] % 8 or x.shape[2] % 8 or x.shape[3] % 8:
            raise ValueError(f"CostRegNet needs D,h,w divisible by 8, got {tuple(x.shape[1:4])}")
        pk = self._prepare()

        def layer(name, t, relu=True, skip=None):
            w, cin, cout, scale, shift, mode = pk[name]
            return ops.conv3d_bn_relu(t, w, cin, cout, scale, shift, mode, relu=relu, skip=skip)

        if x.dtype == torch.float16:  # the fp16 volume of ops.warp_variance_f16: first layer on fp16 MFMA, fp32 out
            _, _, _, scale0, shift0, _ = pk["conv0"]
            conv0 = ops.conv3d_bn_relu_f16in(x, pk["conv0_f16"], scale0, shift0, relu=True)
            a0 = None
        elif self.conv0_split:
            _, _, _, scale0, shift0, _ = pk["conv0"]
            # The range of conv0's output for conv1: an upper BOUND from max |x| and the layer's weights instead of a measurement
            # (|y_c| <= |scale_c| sum|w_c| max|x| + |shift_c|).  Measuring costs a pass over 453 MB (85 us) or, as a by-product of the
            # kernel's store epilogue, 58 us; the bound is 10-50x loose, which moves the absolute error floor of conv1's activations
            # from 2^-39 to about 2^-33 of the true maximum: still below fp32's own rounding of the sums they enter.
            if x_absmax is None:
                x_absmax = ops.absmax(x)
            conv0 = ops.conv3d_bn_relu_split(x, pk["conv0_split"], scale0, shift0, relu=True, x_absmax=x_absmax)
            gain0, off0 = pk["conv0_bound"]
            a0 = (gain0 * x_absmax + off0).amax().reshape(1)
        else:
            conv0 = layer("conv0", x)
        if self.conv0_split:
            # conv1 .. conv6 on the implicit-GEMM split kernel of the 2-D engine (depth taps as chunks), each scaled by max |x| of its
            # input, which the layer before leaves behind as a by-product of its store epilogue; the transposed layers and `prob`
            # on the fp32 matrix instruction / vector ALU (measured faster there: tools/bench_k4_igemm.py)
            slots = torch.zeros(8, dtype=torch.float32, device=x.device)  # the layers' max-|y| slots: one fill for all
            nslot = [0]

            def ig(name, t, a, want_absmax=True):
                _, cin, cout, sc, sh, mode = pk[name]
                out_a = None
                if want_absmax:
                    out_a = slots[nslot[0]:nslot[0] + 1]
                    nslot[0] += 1
                return ops.conv3d_bn_relu_igemm(t, a, pk[name + "_igemm"], cin, cout, sc, sh, mode, relu=True, return_absmax=want_absmax,
                                                out_absmax=out_a)

            if a0 is None:
                a0 = ops.absmax(conv0)
            t, a = ig("conv1", conv0, a0)
            conv2, a = ig("conv2", t, a)
            t, a = ig("conv3", conv2, a)
            conv4, a = ig("conv4", t, a)
            t, a = ig("conv5", conv4, a)
            y = ig("conv6", t, a, want_absmax=False)
        else:
            conv2 = layer("conv2", layer("conv1", conv0))
            conv4 = layer("conv4", layer("conv3", conv2))
            y = layer("conv6", layer("conv5", conv4))
        y = layer("conv7", y, skip=conv4)
        del conv4
        y = layer("conv9", y, skip=conv2)
        del conv2
        y = layer("conv11", y, skip=conv0)
        del conv0
        return layer("prob", y, relu=False).squeeze(-1)

    @ops.inference_only
    def forward(self, x):
        """Reference layout: (B,32,D,h,w) -> (B,1,D,h,w)."""
        return self.forward_channels_last(ops.to_channels_last_3d(x)).unsqueeze(1)


# ------------------------------------------------------------------------------------------------
# DispNet 2-D CNN around the Path-A sweep (plain torch; MIOpen)
# ------------------------------------------------------------------------------------------------
class _ConvLeaky(nn.Sequential):
    """[0] = Conv2d / ConvTranspose2d with bias, [1] = LeakyReLU(0.2) — the reference's nn.Sequential (same state-dict
    keys).  Inference on the GPU runs the convolution without bias on the vendor library and applies bias + LeakyReLU
    in one in-place pass (ops.bias_leaky_relu_) instead of torch's two; same fp32 operations, bit-identical."""

    def forward(self, x):
        c = self[0]
        if torch.is_grad_enabled() or not x.is_cuda or c.bias is None:
            return super().forward(x)
        if isinstance(c, nn.ConvTranspose2d):
            y = F.conv_transpose2d(x, c.weight, None, c.stride, c.padding, c.output_padding, c.groups, c.dilation)
        else:
            y = F.conv2d(x, c.weight, None, c.stride, c.padding, c.dilation, c.groups)
        if y.dtype != torch.float32 or not y.is_contiguous() or y.shape[0] * y.shape[1] > 65535:  # (fp16: the autocast variant)
            return F.leaky_relu(y + c.bias.to(y.dtype).view(1, -1, 1, 1), self[1].negative_slope, inplace=True)
        return ops.bias_leaky_relu_(y, c.bias, self[1].negative_slope)


def conv(cin, cout, kernel_size=3, stride=1):
    return _ConvLeaky(nn.Conv2d(cin, cout, kernel_size=kernel_size, stride=stride, padding=(kernel_size - 1) // 2, bias=True),
                      nn.LeakyReLU(0.2, inplace=True))


class ReLUAndSigmoid(nn.Module):
    """Channel 0: ReLU (inverse depth); other channels: range-scaled sigmoid (log scale), blocks/utils.py:30-41."""

    def __init__(self, inplace=False, min=0.0, max=1.0):
        super().__init__()
        self.min, self.max, self.range = min, max, max - min

    def forward(self, x):
        return torch.cat([F.relu(x[:, :1]), torch.sigmoid(x[:, 1:] * (4 / self.range)) * self.range + self.min], 1)


class DispnetEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = conv(3, 64, kernel_size=7, stride=2)
        self.conv2 = conv(64, 128, kernel_size=5, stride=2)
        self.conv3 = conv(128, 256, kernel_size=3, stride=2)

    def forward(self, image):
        c1 = self.conv1(image)
        c2 = self.conv2(c1)
        c3 = self.conv3(c2)
        return {"conv1": c1, "conv2": c2, "conv3a": c3}, c3


class DispnetContextEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv_redir = conv(256, 32, kernel_size=1, stride=1)

    def forward(self, conv3):
        return self.conv_redir(conv3)


class DispnetCostvolumeEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv3_1 = conv(256 + 32, 256)
        self.conv4 = conv(256, 512, stride=2)
        self.conv4_1 = conv(512, 512)
        self.conv5 = conv(512, 512, stride=2)
        self.conv5_1 = conv(512, 512)
        self.conv6 = conv(512, 1024, stride=2)
        self.conv6_1 = conv(1024, 1024)

    def forward(self, corr, ctx):
        out = {"merged": torch.cat([ctx, corr], 1)}
        x = out["merged"]
        for name in ("conv3_1", "conv4", "conv4_1", "conv5", "conv5_1", "conv6", "conv6_1"):
            x = getattr(self, name)(x)
            out[name] = x
        return out, x


def _pred_block(cin):
    return nn.Sequential(nn.Conv2d(cin, 2, kernel_size=3, stride=1, padding=1, bias=True),
                         ReLUAndSigmoid(inplace=True, min=-10, max=10))


def _deconv(cin, cout):
    return _ConvLeaky(nn.ConvTranspose2d(cin, cout, kernel_size=4, stride=2, padding=1, bias=True),
                      nn.LeakyReLU(0.2, inplace=True))


def _iconv(cin, cout):
    return _ConvLeaky(nn.Conv2d(cin + 2, cout, kernel_size=3, stride=1, padding=1, bias=True),
                      nn.LeakyReLU(0.2, inplace=True))


class DispnetDecoder(nn.Module):
    """5-level refinement decoder with an (inverse depth, log b) head per level (dispnet_decoder.py:36-138)."""

    SKIPS = ["conv5_1", "conv4_1", "conv3_1", "conv2", "conv1"]
    SKIP_CH = [512, 512, 256, 128, 64]

    def __init__(self):
        super().__init__()
        ch = 1024
        self.pred_0 = _pred_block(ch)
        for lvl, skip_ch in enumerate(self.SKIP_CH, start=1):
            nxt = ch // 2
            setattr(self, f"deconv_{lvl}", _deconv(ch, nxt))
            setattr(self, f"rfeat{lvl}", _iconv(nxt + skip_ch, nxt))
            setattr(self, f"pred_{lvl}", _pred_block(nxt))
            ch = nxt

    @staticmethod
    def _record(pred, preds):
        mean, log_b = pred[:, 0:1], pred[:, 1:2]
        ent = torch.log(2 * torch.exp(log_b) + 1e-4) + 1
        preds.setdefault("invdepth_uncertainties_all", []).append(ent)
        preds.setdefault("invdepth_log_bs_all", []).append(log_b)
        preds.setdefault("invdepths_all", []).append(mean)
        preds["invdepth_uncertainty"], preds["invdepth_log_b"], preds["invdepth"] = ent, log_b, mean

    def _head(self, lvl, feat, preds):
        """pred_k block + _record.  At inference on the GPU in fp32 the activation and the entropy are ONE engine launch
        (ops.dispnet_head) instead of torch's ~10 elementwise kernels per head; same formulas."""
        blk = getattr(self, f"pred_{lvl}")
        if torch.is_grad_enabled() or not feat.is_cuda or feat.dtype != torch.float32:
            pred = blk(feat)
            self._record(pred, preds)
            return pred
        pred, ent = ops.dispnet_head(blk[0](feat))
        mean, log_b = pred[:, 0:1], pred[:, 1:2]
        preds.setdefault("invdepth_uncertainties_all", []).append(ent)
        preds.setdefault("invdepth_log_bs_all", []).append(log_b)
        preds.setdefault("invdepths_all", []).append(mean)
        preds["invdepth_uncertainty"], preds["invdepth_log_b"], preds["invdepth"] = ent, log_b, mean
        return pred

    def forward(self, enc_fused, all_enc):
        preds = {}
        feat, pred = enc_fused, self._head(0, enc_fused, preds)
        for lvl, skip in enumerate(self.SKIPS, start=1):
            up = getattr(self, f"deconv_{lvl}")(feat)
            pred_up = F.interpolate(pred, size=up.shape[-2:], mode="bilinear", align_corners=False).detach()
            feat = getattr(self, f"rfeat{lvl}")(torch.cat((all_enc[skip], up, pred_up), 1))
            pred = self._head(lvl, feat, preds)
        return preds
