#!/usr/bin/env python3
"""Per-layer timing of FeatureNet on the HIP conv2d kernels (K6) at a BASELINE config, next to the same layers as
torch modules on MIOpen.  GPU box only:  python tools/bench_k6.py [--config 2] [--iters 10]"""
import argparse, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robustmvd_amd import ops, _lib as L
import robustmvd_amd as R

CONFIGS = {1: (448, 640, 2, 128), 2: (768, 1152, 4, 256), 3: (896, 1216, 4, 256), 4: (704, 1280, 6, 512)}
ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=2); ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--no-miopen", action="store_true")
args = ap.parse_args()
H, W, V, D = CONFIGS[args.config]
N = V + 1
dev = torch.device("cuda:0")
net = R.blocks.FeatureNet().eval().to(dev)
pk = net._prepare()


def timed(fn):
    for _ in range(2): y = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters): y = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters, y


x = torch.rand(N, 3, H, W, device=dev)
tot = tot_mi = 0.0
xm = x
for i, (w, cin, cout, k, st, sc, sh, relu) in enumerate(pk):
    ms, y = timed(lambda: ops.conv2d_bn_relu(x, w, cin, cout, k, st, sc, sh, relu=relu))
    gf = y.numel() * cin * k * k * 2 / 1e9
    mb = (x.numel() + y.numel()) * 4 / 1e6
    line = f"layer {i}: {cin:2d}->{cout:2d} k{k} s{st} out {tuple(y.shape[1:3])}: {ms:7.3f} ms  {gf/ms:8.1f} TFLOP/s  {mb/ms:7.1f} GB/s"
    if not args.no_miopen:
        mod = getattr(net, f"conv{i}") if i < 7 else net.feature
        with torch.no_grad():
            ms_mi, ym = timed(lambda: mod(xm))
        line += f"   | torch/MIOpen {ms_mi:7.3f} ms"
        tot_mi += ms_mi
        xm = ym
    print(line, flush=True)
    tot += ms
    x = y
print(f"total K6 {tot:.3f} ms" + ("" if args.no_miopen else f"   torch/MIOpen {tot_mi:.3f} ms"))
with torch.no_grad():
    xi = torch.rand(N, 3, H, W, device=dev)
    ms, _ = timed(lambda: net.forward_layout(xi, L.LAYOUT_NHWC))
print(f"FeatureNet.forward_layout (8 launches back to back): {ms:.3f} ms")
