#!/usr/bin/env python3
"""K4's layers at a BASELINE config (default configs[2]: 256 planes of 192x288): the split-operand implicit-GEMM kernel
(ops.conv3d_bn_relu_igemm) beside the fp32-MFMA kernel (ops.conv3d_bn_relu): us per layer and the difference.  GPU box only."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robustmvd_amd import ops, _lib as L
if os.environ.get("MVD_ALT_LIB"):
    L.use_experiments_library(os.environ["MVD_ALT_LIB"]).__enter__()
ONLY = os.environ.get("MVD_K4_ONLY", "").split(",") if os.environ.get("MVD_K4_ONLY") else None
D, h, w = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 192, 288)
dev = torch.device("cuda:0")
LAYERS = [("conv1", 1, 8, 16, 1), ("conv3", 1, 16, 32, 2), ("conv5", 1, 32, 64, 4), ("conv6", 0, 64, 64, 8), ("conv7", 2, 64, 32, 8),
          ("conv9", 2, 32, 16, 4), ("conv11", 2, 16, 8, 2), ("conv2", 0, 16, 16, 2), ("conv4", 0, 32, 32, 4)]


def timeit(fn, n=10):
    for _ in range(3):
        y = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        y = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, y


for name, mode, cin, cout, div in LAYERS:
    if ONLY and name not in ONLY:
        continue
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.rand(1, D // div, h // div, w // div, cin, generator=g).to(dev)
    wshape = (cin, cout, 3, 3, 3) if mode == 2 else (cout, cin, 3, 3, 3)
    wt = (torch.randn(*wshape, generator=g) * (2.0 / (cin * 27)) ** 0.5).to(dev)
    sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), (torch.randn(cout, generator=g) * 0.1).to(dev)
    w32, _, _ = ops.pack_conv3d_weights(wt, mode)
    wig = ops.pack_conv3d_weights_igemm(wt, mode)
    skip = torch.rand(1, D // div * 2, h // div * 2, w // div * 2, cout, generator=g).to(dev) if mode == 2 else None
    am = ops.absmax(x)
    t0, y0 = timeit(lambda: ops.conv3d_bn_relu(x, w32, cin, cout, sc, sh, mode, relu=True, skip=skip))
    t1, y1 = timeit(lambda: ops.conv3d_bn_relu_igemm(x, am, wig, cin, cout, sc, sh, mode, relu=True, skip=skip, return_absmax=True)[0])
    print(f"{name:7s} mode {mode} {cin:3d}->{cout:3d} in {D // div}x{h // div}x{w // div}: fp32 MFMA {t0:7.1f} us, igemm split {t1:7.1f} us  x{t0 / t1:4.2f}  "
          f"rel diff {float((y1 - y0).abs().max()) / float(y0.abs().max()):.1e}", flush=True)
