#!/usr/bin/env python3
"""Per-layer timing of the CostRegNet kernels (K4) at a BASELINE config. GPU box only."""
import argparse, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from robustmvd_amd import ops, _lib as L
import robustmvd_amd as R

CONFIGS = {1: (448, 640, 2, 128), 2: (768, 1152, 4, 256), 3: (896, 1216, 4, 256), 4: (704, 1280, 6, 512)}
ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=2); ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--only", default="")
ap.add_argument("--exp", action="store_true", help="route through lib_exp/libmvd_hip_exp.so (MVD_K4_* selectors apply)")
args = ap.parse_args()
if os.environ.get("MVD_ALT_LIB"):
    L.use_experiments_library(os.environ["MVD_ALT_LIB"]).__enter__()
elif args.exp:
    L.use_experiments_library().__enter__()
H, W, V, D = CONFIGS[args.config]
h, w = H // 4, W // 4
dev = torch.device("cuda:0")
net = R.CostRegNet().eval().to(dev)
pk = net._prepare()
def rnd(*s): return torch.randn(*s, device=dev)
shapes = {"conv0": (D, h, w, 32), "conv1": (D, h, w, 8), "conv2": (D//2, h//2, w//2, 16), "conv3": (D//2, h//2, w//2, 16),
          "conv4": (D//4, h//4, w//4, 32), "conv5": (D//4, h//4, w//4, 32), "conv6": (D//8, h//8, w//8, 64),
          "conv7": (D//8, h//8, w//8, 64), "conv9": (D//4, h//4, w//4, 32), "conv11": (D//2, h//2, w//2, 16), "prob": (D, h, w, 8)}
flops = {}
tot = 0.0
for name, shp in shapes.items():
    if args.only and name not in args.only.split(","): continue
    wgt, cin, cout, sc, sh, mode = pk[name]
    x = rnd(1, *shp)
    skip = None
    if mode == L.DECONV3D_STRIDE2:
        skip = rnd(1, shp[0]*2, shp[1]*2, shp[2]*2, cout)
    for _ in range(2): y = ops.conv3d_bn_relu(x, wgt, cin, cout, sc, sh, mode, relu=(name != "prob"), skip=skip)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters): y = ops.conv3d_bn_relu(x, wgt, cin, cout, sc, sh, mode, relu=(name != "prob"), skip=skip)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    nvox_out = y.numel() // cout
    taps = 27 if mode != L.DECONV3D_STRIDE2 else 27 / 8
    gf = nvox_out * taps * cin * cout * 2 / 1e9
    tot += ms
    print(f"{name:7s} {cin:3d}->{cout:2d} mode {mode} in {tuple(shp)}: {ms:7.3f} ms  {gf/ms:8.1f} useful GFLOP/s  ({gf:.1f} GFLOP)", flush=True)
    del x, y, skip
print(f"total {tot:.3f} ms")
