#!/usr/bin/env python3
"""FeatureNet's layers (mvsnet_components.py:8-22) at a BASELINE config on the split-operand implicit-GEMM kernel (ops.conv2d_split)
beside the library's fp32-MFMA conv2d kernels (ops.conv2d_bn_relu): us per layer.  GPU box only."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robustmvd_amd import ops, _lib as L
H, W, V = 768, 1152, 4
dev = torch.device("cuda:0")
LAYERS = [("conv1 8->8", 3, 1, 8, 8, H, W), ("conv2 8->16 5x5s2", 5, 2, 8, 16, H, W), ("conv3 16->16", 3, 1, 16, 16, H // 2, W // 2),
          ("conv5 16->32 5x5s2", 5, 2, 16, 32, H // 2, W // 2), ("conv6 32->32", 3, 1, 32, 32, H // 4, W // 4)]


def timeit(fn, n=10):
    for _ in range(3):
        y = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        y = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, y


for name, k, stride, cin, cout, h, w in LAYERS:
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.rand(V + 1, h, w, cin, generator=g).to(dev)
    wt = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(dev)
    sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), (torch.randn(cout, generator=g) * 0.1).to(dev)
    wts = ops.pack_conv2d_weights_split(wt * sc.view(-1, 1, 1, 1), sh, stride=stride)   # BN scale folded into the weights
    am = ops.absmax(x)
    yam = torch.zeros(1, device=dev)
    te, ye = timeit(lambda: ops.conv2d_split(x, am, wts, act=2, out_absmax=yam))
    pk = ops.pack_conv2d_weights(wt)
    tl, yl = timeit(lambda: ops.conv2d_bn_relu(x, pk[0], cin, cout, k, stride, sc, sh, relu=True, out_layout=L.LAYOUT_NHWC))
    d = float((ye - yl).abs().max()) / float(yl.abs().max())
    print(f"{name:22s} {V + 1}x{h}x{w}: engine {te:7.1f} us   fp32-MFMA kernel {tl:7.1f} us   x{tl / te:4.2f}   rel diff {d:.1e}", flush=True)
