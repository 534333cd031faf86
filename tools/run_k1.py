#!/usr/bin/env python3
"""Runs the sweep-correlation op (K1) a few times at the Path-A shape of a BASELINE config (for rocprofv3 passes)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_common as gc
import robustmvd_amd as R
CONFIGS = {1: (448, 640, 2), 2: (768, 1152, 4), 3: (896, 1216, 4)}
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H, W, V = CONFIGS[cfg]
h, w = H // 8, W // 8
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
fk = torch.randn(1, 256, h, w, device=dev)
fs = [torch.randn(1, 256, h, w, device=dev) for _ in range(V)]
K = gc.synthetic_intrinsics(H, W) / np.array([[W] * 3, [H] * 3, [1.0] * 3], np.float32)
Ts = [T(gc.synthetic_pose(rng)[None]) for _ in range(V)]
blk = R.PlanesweepCorrelation()
with torch.no_grad():
    for _ in range(n):
        blk(fk, T(K[None].astype(np.float32)), fs, Ts, num_sampling_points=256, min_depth=0.4, max_depth=1000.0)
torch.cuda.synchronize()
