#!/usr/bin/env python3
"""Throughput of MVSNet.forward with N frames in flight (FramePipeline) at the headline config.  GPU box only."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench as BN
import robustmvd_amd as R
dev = torch.device("cuda:0")
H, W, V, D = BN.CONFIGS[2]
model, _ = BN.build_mvsnet(D, dev)
samples = [BN.adapted_sample(model, f, H, W, V, (np.float32(0.5), np.float32(10.0))) for f in range(2)]
with torch.no_grad():
    for _ in range(10):
        model(**samples[0])
torch.cuda.synchronize()
for depth in (1, 2, 3, 4):
    pipe = R.FramePipeline(model, depth=depth)
    for i in range(8):
        pipe.submit(**samples[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for i in range(n):
        pipe.submit(**samples[i % 2])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"frames in flight {depth}: {n / dt:.1f} maps/s ({dt / n * 1e3:.3f} ms per map)", flush=True)
