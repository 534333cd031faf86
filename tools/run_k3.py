#!/usr/bin/env python3
"""Runs the warp+variance op a few times at a BASELINE config (for rocprofv3 passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_k3 import make_inputs, CONFIGS
from robustmvd_amd import ops
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H, W, V, D = CONFIGS[cfg]
feats, projs, key_inv, depth = make_inputs(H, W, V, D, torch.device("cuda:0"))
for _ in range(n):
    out = ops.warp_variance(feats[0], feats[1:], projs, key_inv, depth, channels_last=True)
torch.cuda.synchronize()
