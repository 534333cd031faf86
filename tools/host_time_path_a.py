#!/usr/bin/env python3
"""Host (CPU) time RobustMVD.forward needs to ENQUEUE one frame, against the GPU time of the frame.  GPU box only."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench as BN
dev = torch.device("cuda:0")
H, W, V, D = BN.CONFIGS[2]
for name, build in (("robust_mvd", BN.build_robustmvd), ("mvsnet", lambda d: BN.build_mvsnet(D, d))):
    model, _ = build(dev)
    s = BN.adapted_sample(model, 0, H, W, V) if name == "robust_mvd" else BN.adapted_sample(model, 0, H, W, V, depth_range=(0.5, 10.0))
    with torch.no_grad():
        for _ in range(5):
            model(**s)
        torch.cuda.synchronize()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            model(**s)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"{name}: host enqueue {1e3 * (t1 - t0) / n:.3f} ms per frame, with the GPU drained {1e3 * (t2 - t0) / n:.3f} ms per frame")
