#!/usr/bin/env python3
"""Runs FeatureNet (K6 x 8) a few times at a BASELINE config (for rocprofv3 passes): run_k6.py [config] [n]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import robustmvd_amd as R
from robustmvd_amd import _lib as L
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
CONFIGS = {1: (448, 640, 2, 128), 2: (768, 1152, 4, 256), 3: (896, 1216, 4, 256), 4: (704, 1280, 6, 512)}
H, W, V, D = CONFIGS[cfg]
dev = torch.device("cuda:0")
net = R.blocks.FeatureNet().eval().to(dev)
x = torch.rand(V + 1, 3, H, W, device=dev)
with torch.no_grad():
    for _ in range(n):
        y = net.forward_layout(x, L.LAYOUT_NHWC_BORDER)
torch.cuda.synchronize()
