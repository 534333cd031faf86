#!/usr/bin/env python3
"""Times the backward (VJP) kernels of the sweep operators at BASELINE configs[1] sizes: K3 (scatter of the
variance gradient into the feature maps with float atomics), K1, K2.  Reports the time of `.backward()` of the autograd wrapper
(kernel + the wrapper's layout copies) and the algorithmic rates next to the guide's ceilings: ~1.3 TB/s of added bytes for
no-return float atomics (MI355X_MICROARCH.md, Global float atomics), ~5-6 TB/s for streams.  GPU box only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen_common as gc  # noqa: E402
from robustmvd_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
ATOMIC_CEILING_GBS = 1300.0


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def k3():
    from test_hip_shapes import mvs_inputs
    B, C, h, w, D, V = 1, 32, 112, 160, 128, 2  # configs[1]: 448x640, 2 sources, 128 planes
    feats, projs, key_inv, depth = mvs_inputs(B, C, h, w, D, V, seed=1)
    T = lambda a: torch.from_numpy(a).to(dev)
    ft = [T(f).requires_grad_(True) for f in feats]
    pr, ki, dv = [T(p) for p in projs], T(key_inv), T(depth)
    G = torch.randn(B, C, D, h, w, device=dev)

    def step():
        for f in ft:
            f.grad = None
        var = ops.warp_variance_autograd(ft[0], ft[1:], pr, ki, dv)
        var.backward(G)

    def fwd_only():
        with torch.no_grad():
            ops.warp_variance(ft[0].detach(), [f.detach() for f in ft[1:]], pr, ki, dv)

    t_all, t_f = timed(step), timed(fwd_only)
    ms = t_all - t_f
    atom = B * D * h * w * V * C * 4 * 4.0      # 4 taps x 4 bytes per (pixel, plane, view, channel)
    print(f"K3 backward  {B}x{C}x{D}x{h}x{w} V{V}: forward+backward {t_all:.3f} ms, forward {t_f:.3f} ms -> backward {ms:.3f} ms; "
          f"{atom / 1e9:.2f} GB of float atomics -> {atom / ms / 1e6:.0f} GB/s = {atom / ms / 1e6 / ATOMIC_CEILING_GBS:.2f} of the "
          f"~{ATOMIC_CEILING_GBS:.0f} GB/s atomic ceiling")


def k1():
    import robustmvd_amd as R
    N, C, h, w, S, V = 1, 256, 56, 80, 256, 2   # configs[1] Path A: 448x640 -> 56x80 features
    rng = np.random.default_rng(2)
    fk = torch.from_numpy(rng.standard_normal((N, C, h, w)).astype(np.float32)).to(dev).requires_grad_(True)
    fs = [torch.from_numpy(rng.standard_normal((N, C, h, w)).astype(np.float32)).to(dev).requires_grad_(True) for _ in range(V)]
    K = torch.tensor([[[0.9, 0, 0.5], [0, 0.9 * 640 / 448, 0.5], [0, 0, 1]]], device=dev)
    Ts = [torch.from_numpy(gc.synthetic_pose(rng)[None]).to(dev) for _ in range(V)]
    blk = R.PlanesweepCorrelation()
    G = [torch.randn(N, S, h, w, device=dev) for _ in range(V)]

    def step():
        fk.grad = None
        for f in fs:
            f.grad = None
        corrs, masks, _ = blk(feat_key=fk, intrinsics_key=K, feat_sources=fs, source_to_key_transforms=Ts, intrinsics_sources=[K] * V,
                              num_sampling_points=S, min_depth=0.4, max_depth=1000.0)
        torch.autograd.backward(corrs, G)

    def fwd_only():
        with torch.no_grad():
            blk(feat_key=fk, intrinsics_key=K, feat_sources=fs, source_to_key_transforms=Ts, intrinsics_sources=[K] * V,
                num_sampling_points=S, min_depth=0.4, max_depth=1000.0)

    t_all, t_f = timed(step), timed(fwd_only)
    ms = t_all - t_f
    flops = V * S * h * w * C * 2 * (4 + 4)    # per (pixel, plane, view, channel): 4 tap FMAs into d(key), 4 into d(source taps)
    print(f"K1 backward  C{C} {h}x{w} S{S} V{V}: forward+backward {t_all:.3f} ms, forward {t_f:.3f} ms -> backward {ms:.3f} ms; "
          f"{flops / 1e9:.1f} GFLOP -> {flops / ms / 1e9:.1f} TFLOP/s (fp32 vector peak 157)")


def k2():
    N, S, h, w, V = 1, 256, 56, 80, 2
    rng = np.random.default_rng(3)
    mk = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(dev)
    corrs = [mk(N, S, h, w).requires_grad_(True) for _ in range(V)]
    masks = [(mk(N, S, h, w) > -0.5).float() for _ in range(V)]
    scores = [mk(N, 1, h, w).requires_grad_(True) for _ in range(V)]
    G = torch.randn(N, S, h, w, device=dev)

    def step():
        for t in corrs + scores:
            t.grad = None
        f, _ = ops.fuse_views_autograd(corrs, masks, scores)
        f.backward(G)

    def fwd_only():
        with torch.no_grad():
            ops.fuse_views([c.detach() for c in corrs], masks, [s.detach() for s in scores])

    t_all, t_f = timed(step), timed(fwd_only)
    ms = t_all - t_f
    nbytes = (3 * V + 1) * N * S * h * w * 4.0   # reads corr, mask (twice: two passes share them through the cache), gfused; writes gcorr
    print(f"K2 backward  S{S} {h}x{w} V{V}: forward+backward {t_all:.3f} ms, forward {t_f:.3f} ms -> backward {ms:.3f} ms; "
          f"{nbytes / 1e6:.0f} MB -> {nbytes / ms / 1e6:.0f} GB/s")


if __name__ == "__main__":
    k3(); k1(); k2()
