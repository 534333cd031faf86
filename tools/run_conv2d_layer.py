#!/usr/bin/env python3
"""Runs ONE 2-D layer on the split-operand engine a few times (for rocprofv3): tools/run_conv2d_layer.py k stride mode cin cout B H W [iters]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robustmvd_amd import ops, _lib as L
if os.environ.get("MVD_ALT_LIB"):
    L.use_experiments_library(os.environ["MVD_ALT_LIB"]).__enter__()
k, stride, mode, cin, cout, B, H, W = [int(a) for a in sys.argv[1:9]]
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 5
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
wshape = (cin, cout, 4, 4) if mode == 1 else (cout, cin, k, k)
wt = (torch.randn(*wshape, generator=g) * 0.05).to(dev)
bias = torch.zeros(cout, device=dev)
wts = ops.pack_conv2d_weights_split(wt, bias, stride=stride, mode=mode)
x = torch.randn(B, 3, H, W, device=dev) if mode == 2 else torch.randn(B, H, W, wts.cin_pad, device=dev)
am = ops.absmax(x)
yam = torch.zeros(1, device=dev)
for _ in range(3 if len(sys.argv) > 10 else 0):
    y = ops.conv2d_split(x, am, wts, out_absmax=yam)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    y = ops.conv2d_split(x, am, wts, out_absmax=yam)
e1.record()
torch.cuda.synchronize()
print("ok", tuple(y.shape), "%.1f us per call" % (e0.elapsed_time(e1) / iters * 1e3))
