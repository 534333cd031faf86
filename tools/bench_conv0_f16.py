#!/usr/bin/env python3
"""Times the fp16-input first regulariser layer (conv0 on fp16 MFMA) at a BASELINE config.  GPU box only."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robustmvd_amd import ops, _lib as L
if os.environ.get("MVD_ALT_LIB"):
    L.use_experiments_library(os.environ["MVD_ALT_LIB"]).__enter__()
CONFIGS = {1: (448, 640, 128), 2: (768, 1152, 256), 3: (896, 1216, 256)}
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H, W, D = CONFIGS[cfg]
h, w = H // 4, W // 4
dev = torch.device("cuda:0")
x = (torch.randn(1, D, h, w, 32, device=dev)).half()
wt = torch.randn(8, 32, 3, 3, 3, device=dev) * 0.05
pk = ops.pack_conv3d_weights_f16(wt)
sc, sh = torch.ones(8, device=dev), torch.zeros(8, device=dev)
for _ in range(3):
    y = ops.conv3d_bn_relu_f16in(x, pk, sc, sh)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    y = ops.conv3d_bn_relu_f16in(x, pk, sc, sh)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
gb = (x.numel() * 2 + y.numel() * 4) / 1e9
print(f"conv0_f16 {D}x{h}x{w}: {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s algorithmic ({gb:.2f} GB), {D*h*w*27*32*8*2/ms/1e9:.0f} TFLOP/s useful; checksum {float(y.double().sum()):.6e}")
