#!/usr/bin/env python3
"""Throughput of the Path B forward with 1, 2 or 3 frames in flight on separate HIP streams of one process.
GPU box only: python tools/two_streams.py [config]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench as BN
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
H, W, V, D = BN.CONFIGS[cfg]
model, sd = BN.build_mvsnet(D, dev)
samples = [BN.adapted_sample(model, f, H, W, V, (np.float32(0.5), np.float32(10.0))) for f in range(3)]
with torch.no_grad():
    for _ in range(12):
        model(**samples[0])
    torch.cuda.synchronize()
    for nstream in (1, 2, 3, 1, 2):
        streams = [torch.cuda.Stream(dev) for _ in range(nstream)]
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            K = 24
            for i in range(K):
                with torch.cuda.stream(streams[i % nstream]):
                    model(**samples[i % 3])
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{nstream} stream(s): {dt / K * 1e3:.3f} ms/frame  {K / dt:.1f} maps/s", flush=True)
