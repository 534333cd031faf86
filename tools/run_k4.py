#!/usr/bin/env python3
"""Runs one CostRegNet layer a few times at a BASELINE config (for rocprofv3 passes): run_k4.py <layer> [config] [n]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import robustmvd_amd as R
from robustmvd_amd import ops, _lib as L
name = sys.argv[1]; cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 2; n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
CONFIGS = {1: (448, 640, 2, 128), 2: (768, 1152, 4, 256), 3: (896, 1216, 4, 256), 4: (704, 1280, 6, 512)}
H, W, V, D = CONFIGS[cfg]; h, w = H // 4, W // 4
shapes = {"conv0": (D, h, w, 32), "conv1": (D, h, w, 8), "conv2": (D//2, h//2, w//2, 16), "conv11": (D//2, h//2, w//2, 16), "prob": (D, h, w, 8)}
dev = torch.device("cuda:0")
net = R.CostRegNet().eval().to(dev); pk = net._prepare()
wgt, cin, cout, sc, sh, mode = pk[name]
x = torch.randn(1, *shapes[name], device=dev)
skip = torch.randn(1, shapes[name][0]*2, shapes[name][1]*2, shapes[name][2]*2, cout, device=dev) if mode == L.DECONV3D_STRIDE2 else None
for _ in range(n):
    y = ops.conv3d_bn_relu(x, wgt, cin, cout, sc, sh, mode, relu=(name != "prob"), skip=skip)
torch.cuda.synchronize()
