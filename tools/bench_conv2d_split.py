#!/usr/bin/env python3
"""Every 2-D layer of robust_mvd's DispNet at a BASELINE config (default configs[2]: 768x1152, 4 source views) on the split-operand
engine (ops.conv2d_split) beside the vendor library's convolution + bias + LeakyReLU (torch, NCHW): time per layer and the
difference.  GPU box only.  usage: tools/bench_conv2d_split.py [H W] [V]"""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robustmvd_amd import ops, _lib as L

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (768, 1152)
V = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
# name, k, stride, mode, cin, cout, batch, input H, input W
h2, w2, h4, w4, h8, w8 = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8
LAYERS = [
    ("enc.conv1 (x%d views)" % (V + 1), 7, 2, 2, 3, 64, V + 1, H, W),
    ("enc.conv2 (x%d)" % (V + 1), 5, 2, 0, 64, 128, V + 1, h2, w2),
    ("enc.conv3 (x%d)" % (V + 1), 3, 2, 0, 128, 256, V + 1, h4, w4),
    ("conv_redir", 1, 1, 0, 256, 32, 1, h8, w8),
    ("fusion score 3x3 (x%d)" % V, 3, 1, 0, 256, 128, V, h8, w8),
    ("fusion score 1x1 (x%d)" % V, 1, 1, 0, 128, 1, V, h8, w8),
    ("conv3_1", 3, 1, 0, 288, 256, 1, h8, w8),
    ("conv4", 3, 2, 0, 256, 512, 1, h8, w8),
    ("conv4_1", 3, 1, 0, 512, 512, 1, h8 // 2, w8 // 2),
    ("conv5", 3, 2, 0, 512, 512, 1, h8 // 2, w8 // 2),
    ("conv5_1", 3, 1, 0, 512, 512, 1, h8 // 4, w8 // 4),
    ("conv6", 3, 2, 0, 512, 1024, 1, h8 // 4, w8 // 4),
    ("conv6_1", 3, 1, 0, 1024, 1024, 1, h8 // 8, w8 // 8),
    ("pred_0", 3, 1, 0, 1024, 2, 1, h8 // 8, w8 // 8),
    ("deconv_1", 4, 2, 1, 1024, 512, 1, h8 // 8, w8 // 8),
    ("rfeat1", 3, 1, 0, 1026, 512, 1, h8 // 4, w8 // 4),
    ("deconv_2", 4, 2, 1, 512, 256, 1, h8 // 4, w8 // 4),
    ("rfeat2", 3, 1, 0, 770, 256, 1, h8 // 2, w8 // 2),
    ("deconv_3", 4, 2, 1, 256, 128, 1, h8 // 2, w8 // 2),
    ("rfeat3", 3, 1, 0, 386, 128, 1, h8, w8),
    ("deconv_4", 4, 2, 1, 128, 64, 1, h8, w8),
    ("rfeat4", 3, 1, 0, 194, 64, 1, h4, w4),
    ("deconv_5", 4, 2, 1, 64, 32, 1, h4, w4),
    ("rfeat5", 3, 1, 0, 98, 32, 1, h2, w2),
    ("pred_5", 3, 1, 0, 32, 2, 1, h2, w2),
]


def timeit(fn, n=10):
    for _ in range(3):
        y = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        y = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, y


tot_e = tot_l = 0.0
for name, k, stride, mode, cin, cout, B, hi, wi in LAYERS:
    g = torch.Generator().manual_seed(cin + cout)
    x = (torch.randn(B, cin, hi, wi, generator=g) * 2).to(dev)
    wshape = (cin, cout, 4, 4) if mode == 1 else (cout, cin, k, k)
    wt = (torch.randn(*wshape, generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(dev)
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    wts = ops.pack_conv2d_weights_split(wt, bias, stride=stride, mode=mode)
    if mode == 2:
        xin = x
    else:
        xin = torch.zeros(B, hi, wi, wts.cin_pad, device=dev)
        xin[..., :cin] = x.permute(0, 2, 3, 1)
    am = ops.absmax(x)
    yam = torch.zeros(1, device=dev)

    def lib():
        y = F.conv_transpose2d(x, wt, None, 2, 1) if mode == 1 else F.conv2d(x, wt, None, stride, k // 2)
        return ops.bias_leaky_relu_(y, bias, 0.2)

    te, ye = timeit(lambda: ops.conv2d_split(xin, am, wts, act=1, slope=0.2, out_absmax=yam))
    tl, yl = timeit(lib)
    flops = 2.0 * yl.numel() * cin * (4 if mode == 1 else k * k)
    d = float((ye.permute(0, 3, 1, 2) - yl).abs().max()) / float(yl.abs().max())
    tot_e += te; tot_l += tl
    print(f"{name:26s} {cin:5d}->{cout:5d} {k}x{k}s{stride} {B}x{hi}x{wi}: engine {te:8.1f} us ({flops / te / 1e6:7.1f} TFLOP/s)  library {tl:8.1f} us  "
          f"x{tl / te:5.2f}  rel diff {d:.1e}", flush=True)
print(f"sum: engine {tot_e:.0f} us, library {tot_l:.0f} us")
