#!/usr/bin/env python3
"""conv0 at a BASELINE config: fp32-MFMA kernel vs the split-operand (2 x fp16 terms, range-scaled) kernel: time and
difference; the standalone max |x| pass timed beside it (inside the model it is a by-product of K3).  GPU box only."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robustmvd_amd import ops, _lib as L
CONFIGS = {1: (448, 640, 128), 2: (768, 1152, 256), 3: (896, 1216, 256)}
if os.environ.get("MVD_ALT_LIB"):
    L.use_experiments_library(os.environ["MVD_ALT_LIB"]).__enter__()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
H, W, D = CONFIGS[cfg]
h, w = H // 4, W // 4
dev = torch.device("cuda:0")
x = torch.rand(1, D, h, w, 32, device=dev) * 2
wt = torch.randn(8, 32, 3, 3, 3, device=dev) * 0.05
sc, sh = torch.rand(8, device=dev) + 0.5, torch.randn(8, device=dev) * 0.1
w32, _, _ = ops.pack_conv3d_weights(wt, L.CONV3D_STRIDE1)
wsp = ops.pack_conv3d_weights_split(wt)


def timeit(fn, n=10):
    for _ in range(3):
        y = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        y = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, y


t0, y0 = timeit(lambda: ops.conv3d_bn_relu(x, w32, 32, 8, sc, sh, L.CONV3D_STRIDE1, relu=True))
amax = ops.absmax(x)
t1, y1 = timeit(lambda: ops.conv3d_bn_relu_split(x, wsp, sc, sh, relu=True, x_absmax=amax))
t2, _ = timeit(lambda: ops.absmax(x))
ref = torch.nn.functional.conv3d(x[0].permute(3, 0, 1, 2)[None, :, :32].double(), wt.double(), padding=1)[0, :, :32] if False else None
d = (y1 - y0).abs()
print(f"conv0 {D}x{h}x{w}: fp32 MFMA {t0:.3f} ms, split fp16x2 {t1:.3f} ms ({t0 / t1:.2f}x), absmax pass {t2:.3f} ms; max |diff| {float(d.max()):.3e} "
      f"(output max {float(y0.abs().max()):.3f}), mean |diff| {float(d.mean()):.3e}")
