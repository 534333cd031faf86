#!/usr/bin/env python3
"""RobustMVD.forward at a BASELINE config: the engine's 2-D CNN (engine_dispnet=True, the default) beside the layer-by-layer form on
the vendor library (engine_dispnet=False): ms per frame and the difference of the predictions.  GPU box only.
usage: tools/path_a_engine.py [config] [iters]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench as BN
import robustmvd_amd as R
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
H, W, V, D = BN.CONFIGS[cfg]
model, _ = BN.build_robustmvd(dev)
s = BN.adapted_sample(model, 0, H, W, V)


def timeit(m, label):
    with torch.no_grad():
        for _ in range(5):
            out = m(**s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = m(**s)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    print(f"{label:34s} {ms:7.3f} ms/frame  ({1e3 / ms:6.1f} maps/s)", flush=True)
    return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in out[1].items()}


assert model.engine_dispnet
a = timeit(model, "engine 2-D CNN (default)")
model._engine.side_stream = False
timeit(model, "engine, key encoder in line")
model._engine.side_stream = True
timeit(model, "engine 2-D CNN (default) again")
ref = R.RobustMVD(engine_dispnet=False).eval().to(dev)
ref.load_state_dict(model.state_dict())
b = timeit(ref, "vendor-library convolutions")
for k in ("invdepth", "invdepth_log_b", "invdepth_uncertainty"):
    d = (a[k] - b[k]).abs()
    print(f"  {k}: max |diff| {float(d.max()):.3e} (max |value| {float(b[k].abs().max()):.3e})")
for i in range(6):
    d = (a["invdepths_all"][i] - b["invdepths_all"][i]).abs()
    print(f"  invdepths_all[{i}]: max |diff| {float(d.max()):.3e}")
