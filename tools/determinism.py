#!/usr/bin/env python3
"""Repeats every hot-path kernel at the headline shape on the same input and checks the outputs are bit-identical from
run to run (catches LDS races and uninitialised reads that a single parity run can miss).  GPU box only."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench as BN
import robustmvd_amd as R
from robustmvd_amd import ops, _lib as L
dev = torch.device("cuda:0")
H, W, V, D = BN.CONFIGS[2]
model, _ = BN.build_mvsnet(D, dev)
s = BN.adapted_sample(model, 0, H, W, V, (np.float32(0.5), np.float32(10.0)))
bad = 0
with torch.no_grad():
    ref = model(**s)[0]
    ref = {k: v.clone() for k, v in ref.items()}
    for i in range(6):
        out = model(**s)[0]
        for k in ref:
            if not torch.equal(out[k], ref[k]):
                bad += 1
                print(f"run {i}: {k} differs, max abs {float((out[k] - ref[k]).abs().max()):.3e}")
    # layer by layer on a fixed volume
    net = model.cost_regularization
    pk = net._prepare()
    x = torch.rand(1, D, H // 4, W // 4, 32, device=dev)
    for name in ("conv0", "conv1"):
        w, cin, cout, sc, sh, mode = pk[name]
        a = ops.conv3d_bn_relu(x, w, cin, cout, sc, sh, mode)
        for i in range(4):
            b = ops.conv3d_bn_relu(x, w, cin, cout, sc, sh, mode)
            if not torch.equal(a, b):
                bad += 1
                print(f"{name} run {i} differs")
        x = a
ma, _ = BN.build_robustmvd(dev)
sa = BN.adapted_sample(ma, 0, H, W, V)
with torch.no_grad():
    ra = ma(**sa)[1]["invdepth"].clone()
    for i in range(3):
        if not torch.equal(ma(**sa)[1]["invdepth"], ra):
            print(f"robust_mvd run {i}: invdepth differs (MIOpen kernels may be non-deterministic)")
print("deterministic" if bad == 0 else f"{bad} mismatches")
# the Path-A kernels alone (K1, K2) on fixed inputs
with torch.no_grad():
    g = torch.Generator(device=dev).manual_seed(1)
    h8, w8 = H // 8, W // 8
    fk = torch.randn(1, 256, h8, w8, device=dev, generator=g)
    fs = [torch.randn(1, 256, h8, w8, device=dev, generator=g) for _ in range(V)]
    cb = ma.corr_block
    def k1():
        return cb(feat_key=fk, intrinsics_key=sa["intrinsics"][0], feat_sources=fs, source_to_key_transforms=sa["poses"][1:],
                  intrinsics_sources=sa["intrinsics"][1:], num_sampling_points=256, min_depth=0.4, max_depth=1000.0)
    c0, m0, _ = k1()
    ok = True
    for i in range(3):
        c, m, _ = k1()
        ok &= all(torch.equal(a, b) for a, b in zip(c, c0)) and all(torch.equal(a, b) for a, b in zip(m, m0))
    scores = [torch.randn(1, 1, h8, w8, device=dev, generator=g) for _ in range(V)]
    f0 = ops.fuse_views(c0, m0, scores)
    for i in range(3):
        f = ops.fuse_views(c0, m0, scores)
        ok &= torch.equal(f[0], f0[0]) and torch.equal(f[1], f0[1])
    print("K1/K2 deterministic" if ok else "K1/K2 NOT deterministic")
