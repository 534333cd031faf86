#!/usr/bin/env python3
"""FeatureNet's two full-resolution layers: one launch (mvd_conv2d_head_f32) against two launches of the fp32-MFMA kernel, at the
headline frame (5 views of 768 x 1152).  GPU box only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robustmvd_amd.blocks import FeatureNet  # noqa: E402
from robustmvd_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, H, W = 5, 768, 1152
net = FeatureNet().eval().to(dev)
pk = net._prepare()
img = torch.randn(B, 3, H, W, device=dev)


def two():
    x = img
    for i in range(2):
        w, cin, cout, k, stride, scale, shift, relu = pk[i]
        x = ops.conv2d_bn_relu(x, w, cin, cout, k, stride, scale, shift, relu=relu)
    return x


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


a, b = two(), ops.conv2d_head(img, *pk[9])
print(f"max |one launch - two launches| = {float((a - b).abs().max()):.3g} (max |y| {float(a.abs().max()):.3g})")
t2, t1 = timed(two), timed(lambda: ops.conv2d_head(img, *pk[9]))
mb = (B * 3 * H * W + B * 8 * H * W) * 4 / 1e6
print(f"two launches {t2:.1f} us, one launch {t1:.1f} us ({mb:.0f} MB algorithmic -> {mb / t1 * 1e3:.0f} GB/s)")
