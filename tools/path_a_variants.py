#!/usr/bin/env python3
"""Times RobustMVD.forward at a BASELINE config in a few execution variants (GPU box):
  default          contiguous NCHW, MIOpen default solver choice
  channels_last    model + inputs in torch.channels_last (NHWC kernels of the vendor library)
  benchmark        torch.backends.cudnn.benchmark = True (MIOpen find mode)
usage: tools/path_a_variants.py [config] [iters]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench as BN
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
H, W, V, D = BN.CONFIGS[cfg]


def timeit(model, s, label):
    with torch.no_grad():
        for _ in range(8):
            out = model(**s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = model(**s)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    print(f"{label:28s} {ms:7.3f} ms/frame  ({1e3 / ms:6.1f} maps/s)", flush=True)
    return out[0]["depth"].clone()


model, _ = BN.build_robustmvd(dev)
s = BN.adapted_sample(model, 0, H, W, V)
ref = timeit(model, s, "default (NCHW)")
torch.backends.cudnn.benchmark = True
d2 = timeit(model, s, "benchmark=True (NCHW)")
print("   max rel diff vs default:", float(((d2 - ref).abs() / ref.abs().clamp_min(1e-6)).max()))
torch.backends.cudnn.benchmark = False
model_cl = model.to(memory_format=torch.channels_last)
s_cl = dict(s)
s_cl["images"] = [im.contiguous(memory_format=torch.channels_last) for im in s["images"]]
d3 = timeit(model_cl, s_cl, "channels_last")
print("   max rel diff vs default:", float(((d3 - ref).abs() / ref.abs().clamp_min(1e-6)).max()))
torch.backends.cudnn.benchmark = True
d4 = timeit(model_cl, s_cl, "channels_last + benchmark")
torch.backends.cudnn.benchmark = False
import robustmvd_amd as R
mh = R.RobustMVD(half_dispnet=True).eval()
mh.load_state_dict(BN.build_robustmvd(torch.device("cpu"))[0].state_dict()) if False else None
mh.load_state_dict({k: v.to("cpu") for k, v in model.state_dict().items()})
mh = mh.to(dev)
d5 = timeit(mh, s, "half_dispnet (fp16 convs)")
print("   max rel diff vs default:", float(((d5 - ref).abs() / ref.abs().clamp_min(1e-6)).max()),
      " median:", float(((d5 - ref).abs() / ref.abs().clamp_min(1e-6)).median()))
