#!/usr/bin/env python3
"""MVSNet.forward at configs[2] with K3's max|volume| by-product (default) and with a constant in its place: what the by-product costs
inside the frame.  GPU box only."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench as BN
from robustmvd_amd import ops
dev = torch.device("cuda:0")
H, W, V, D = BN.CONFIGS[2]
model, _ = BN.build_mvsnet(D, dev)
s = BN.adapted_sample(model, 0, H, W, V, (np.float32(0.5), np.float32(10.0)))


def timeit(label, n=30):
    with torch.no_grad():
        for _ in range(5):
            model(**s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            model(**s)
        torch.cuda.synchronize()
    print(f"{label:40s} {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step", flush=True)


timeit("K3 with the max|x| by-product")
orig = ops.warp_variance
with torch.no_grad():
    ref_amax = orig(*[None] * 0) if False else None
const = torch.full((1,), 4.0, device=dev)


def patched(*a, return_absmax=False, **k):
    out = orig(*a, return_absmax=False, **k)
    return (out, const) if return_absmax else out


ops.warp_variance = patched
timeit("K3 without it (constant range)")
ops.warp_variance = orig
timeit("K3 with the max|x| by-product (again)")
