#!/usr/bin/env python3
"""RobustMVD.forward at configs[2] with K1's max|corr| by-product (default) and with a separate pass over the volumes in its place.
GPU box only."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench as BN
from robustmvd_amd import ops
dev = torch.device("cuda:0")
H, W, V, D = BN.CONFIGS[2]
model, _ = BN.build_robustmvd(dev)
s = BN.adapted_sample(model, 0, H, W, V)


def timeit(label, n=30):
    with torch.no_grad():
        for _ in range(5):
            model(**s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            model(**s)
        torch.cuda.synchronize()
    print(f"{label:44s} {(time.perf_counter() - t0) / n * 1e3:.3f} ms/frame", flush=True)


timeit("K1 with the max|corr| by-product")
orig = ops.sweep_corr_nhwc


def patched(*a, corr_absmax=None, **k):
    out = orig(*a, corr_absmax=None, **k)
    if corr_absmax is not None:
        c = a[6]
        big = torch.as_strided(c[0], (len(c) * c[0].shape[0],) + tuple(c[0].shape[1:]), c[0].stride())
        torch.maximum(corr_absmax, ops.absmax(big), out=corr_absmax)
    return out


ops.sweep_corr_nhwc = patched
timeit("K1 without it + a pass over the volumes")
ops.sweep_corr_nhwc = orig
timeit("K1 with the max|corr| by-product (again)")
