#!/usr/bin/env python3
"""Times the generic sweep reductions (mvd_sweep_reduce_f32: CVP-MVSNet's per-pixel-hypothesis variance, Vis-MVSNet's group-wise
correlation) at BASELINE configs[1] sizes and reports their algorithmic GB/s.  GPU only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from robustmvd_amd import _lib as L, sweep_modes as SM  # noqa: E402

dev = torch.device("cuda:0")
B, C, h, w, D, V = 1, 32, 112, 160, 128, 2


def timed(fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def main():
    from test_hip_shapes import mvs_inputs
    feats, projs, key_inv, depth = mvs_inputs(B, C, h, w, D, V, seed=3)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ft = [T(f) for f in feats]
    Ms = [T((p @ key_inv)[:, :3, :4].astype(np.float32)) for p in projs]
    dv = T(depth)
    dpp = (dv[:, :, None, None] * (1 + 0.01 * torch.rand(B, D, h, w, device=dev))).contiguous()
    in_bytes = (V + 1) * C * h * w * 4.0
    cases = [("variance, shared planes", lambda: SM.sweep_reduce(ft[0], ft[1:], Ms, dv, L.REDUCE_VARIANCE), C * D * h * w * 4.0),
             ("variance, per-pixel hypotheses (cvp)", lambda: SM.sweep_reduce(ft[0], ft[1:], Ms, dpp, L.REDUCE_VARIANCE), C * D * h * w * 4.0 + D * h * w * 4.0),
             ("variance with the key-squared aliasing (cvp quirk)", lambda: SM.sweep_reduce(ft[0], ft[1:], Ms, dpp, L.REDUCE_VARIANCE_KEYSQ), C * D * h * w * 4.0 + D * h * w * 4.0),
             ("group-wise correlation, 8 groups (vis)", lambda: SM.sweep_reduce(ft[0], ft[1:], Ms, dv, L.REDUCE_GROUPCORR, groups=8, pix_offset=0.5, stretch=False), V * 8 * D * h * w * 4.0)]
    for name, fn, out_bytes in cases:
        ms = timed(fn)
        nb = in_bytes + out_bytes
        print(f"sweep_reduce {name:55s} {C}x{D}x{h}x{w} V{V}: {ms:.3f} ms, {nb / 1e6:.0f} MB algorithmic -> {nb / ms / 1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
