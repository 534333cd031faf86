#!/usr/bin/env python3
"""Runs RobustMVD.forward a few times at a BASELINE config (for rocprofv3 --kernel-trace): run_path_a.py [config] [n]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench as BN
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
H, W, V, D = BN.CONFIGS[cfg]
model, _ = BN.build_robustmvd(dev)
s = BN.adapted_sample(model, 0, H, W, V)
with torch.no_grad():
    for _ in range(n):
        model(**s)
torch.cuda.synchronize()
