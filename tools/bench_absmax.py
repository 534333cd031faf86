#!/usr/bin/env python3
"""mvd_absmax_f32 on a few sizes: us and GB/s.  GPU box only."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robustmvd_amd import ops
dev = torch.device("cuda:0")
for mb in (10, 42, 57, 113, 452):
    x = torch.randn(mb * 262144, device=dev)
    for _ in range(3):
        a = ops.absmax(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        a = ops.absmax(x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    assert float(a) == float(x.abs().max())
    print(f"{mb:4d} MiB: {us:6.1f} us  {x.numel() * 4 / us / 1e3:7.1f} GB/s")
