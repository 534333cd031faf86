#!/usr/bin/env python3
"""Micro-benchmark of the warp+variance kernel (K3) at a BASELINE config; sweeps the compiled variants
selected by MVD_K3_CFG.  GPU box only:  python tools/bench_k3.py [--config 2] [--cfgs 4,4 8,2 ...]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_common as gc  # noqa: E402
from robustmvd_amd import ops, _lib as L  # noqa: E402

CONFIGS = {1: (448, 640, 2, 128), 2: (768, 1152, 4, 256), 3: (896, 1216, 4, 256), 4: (704, 1280, 6, 512)}


def make_inputs(H, W, V, D, dev, seed=0):
    h, w, C = H // 4, W // 4, 32
    rng = np.random.default_rng(seed)
    feats = [torch.from_numpy(rng.standard_normal((1, C, h, w)).astype(np.float32)).to(dev) for _ in range(V + 1)]
    K = gc.synthetic_intrinsics(H, W)
    Ks = K.copy()
    Ks[:2] *= 0.25

    def proj(T, key):
        P = T.copy()
        P[:3, :4] = Ks @ P[:3, :4]
        return torch.from_numpy((np.linalg.inv(P) if key else P).astype(np.float32)[None]).to(dev)

    key_inv = proj(np.eye(4, dtype=np.float32), True)
    projs = [proj(gc.synthetic_pose(rng), False) for _ in range(V)]
    depth = torch.linspace(0.5, 10.0, D, device=dev)[None]
    return feats, projs, key_inv, depth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--cfgs", nargs="*", default=["product"])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--layout", default="ndhwc")
    ap.add_argument("--exact", action="store_true", help="exact_grid=True: the reference's rounding chain")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    H, W, V, D = CONFIGS[args.config]
    feats, projs, key_inv, depth = make_inputs(H, W, V, D, dev)
    nbytes = 4.0 * ((V + 1) * 32 * (H // 4) * (W // 4) + 32 * D * (H // 4) * (W // 4))
    ref = None
    cl = args.layout == "ndhwc"
    import contextlib
    for cfg in args.cfgs:
      # "product" = the shipped library (fixed dispatch); anything else = a variant of the experiments library
      # "lib:<path>" = an alternate library (knock-out builds of tools/ko_k3.sh)
      alt = cfg[4:] if cfg.startswith("lib:") else None
      with (contextlib.nullcontext() if cfg == "product" else L.use_experiments_library(alt)):
        os.environ["MVD_K3_CFG"] = "" if alt else cfg
        lib = L.load()
        for _ in range(3):
            out = ops.warp_variance(feats[0], feats[1:], projs, key_inv, depth, channels_last=cl, exact_grid=args.exact)
        ts = []
        for _ in range(args.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e1.record()
            lib.mvd_arm_kernel_timing(e0.cuda_event, e1.cuda_event)
            out = ops.warp_variance(feats[0], feats[1:], projs, key_inv, depth, channels_last=cl, exact_grid=args.exact)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts))
        if ref is None:
            ref = out.clone()
            diff = 0.0
        else:
            diff = float((out - ref).abs().max())
        print(f"cfg {os.path.basename(cfg):>5s} layout {args.layout}: {ms:.3f} ms  {nbytes / ms / 1e6:.0f} GB/s  (min {min(ts):.3f})  maxdiff_vs_first {diff:.2e}",
              flush=True)
        del out


if __name__ == "__main__":
    main()
