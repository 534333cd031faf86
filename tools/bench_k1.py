#!/usr/bin/env python3
"""Times the sweep-correlation kernel (K1) alone at the Path-A shape of a BASELINE config (HIP events around the main
kernel, mvd_arm_kernel_timing).  GPU box only:  python tools/bench_k1.py [config] [iters]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_common as gc
import robustmvd_amd as R
from robustmvd_amd import _lib as L
if os.environ.get("MVD_ALT_LIB"):
    L.use_experiments_library(os.environ["MVD_ALT_LIB"]).__enter__()
CONFIGS = {1: (448, 640, 2), 2: (768, 1152, 4), 3: (896, 1216, 4)}
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
H, W, V = CONFIGS[cfg]
h, w = H // 8, W // 8
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
fk = torch.randn(1, 256, h, w, device=dev)
fs = [torch.randn(1, 256, h, w, device=dev) for _ in range(V)]
K = T((gc.synthetic_intrinsics(H, W) / np.array([[W] * 3, [H] * 3, [1.0] * 3], np.float32))[None].astype(np.float32))
Ts = [T(gc.synthetic_pose(rng)[None]) for _ in range(V)]
blk = R.PlanesweepCorrelation()
lib = L.load()
ts = []
with torch.no_grad():
    for i in range(iters + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()
        lib.mvd_arm_kernel_timing(e0.cuda_event, e1.cuda_event)
        c, m, _ = blk(fk, K, fs, Ts, num_sampling_points=256, min_depth=0.4, max_depth=1000.0)
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(e0.elapsed_time(e1))
flops = V * 256 * h * w * 256 * 10
print(f"K1 {h}x{w} C256 S256 V{V}: median {np.median(ts):.3f} ms (min {min(ts):.3f})  {flops / np.median(ts) / 1e9:.1f} TFLOP/s direct-form; "
      f"checksum {float(sum(x.double().sum() for x in c)):.6e} mask {float(sum(x.sum() for x in m)):.0f}")
