#!/usr/bin/env python3
"""Per-step wall times of the Path B forward at a BASELINE config, plus a stage breakdown with device events."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench as BN
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.backends.cudnn.benchmark = os.environ.get("MVD_BENCH_MIOPEN_FIND", "1") == "1"
dev = torch.device("cuda:0")
H, W, V, D = BN.CONFIGS[cfg]
model, sd = BN.build_mvsnet(D, dev)
samples = [BN.adapted_sample(model, f, H, W, V, (np.float32(0.5), np.float32(10.0))) for f in range(2)]
ts = []
with torch.no_grad():
    for i in range(16):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model(**samples[i % 2])
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("per-step ms:", " ".join(f"{t:.2f}" for t in ts))
# stage breakdown
from robustmvd_amd import ops
s = samples[0]
with torch.no_grad():
    def ev(): e = torch.cuda.Event(enable_timing=True); e.record(); return e
    for rep in range(2):
        e0 = ev()
        n = 1
        depth_samples = model.depth_samples(s["depth_range"], n, dev)
        proj = model.projection_matrices(s["intrinsics"], s["poses"], [0], dev)
        e1 = ev()
        from robustmvd_amd import _lib as L
        feats = list(torch.split(model.feature.forward_layout(torch.cat(s["images"], 0), L.LAYOUT_NHWC_BORDER), n, 0))
        e2 = ev()
        var = ops.warp_variance(feats[0], feats[1:], proj[1:], proj[0], depth_samples, channels_last=True, staged=True)
        e3 = ev()
        cost = model.cost_regularization.forward_channels_last(var)
        e4 = ev()
        depth, conf = ops.softmax_regress(cost, depth_samples)
        e5 = ev()
        torch.cuda.synchronize()
    print("stages ms: prep %.2f  featurenet %.2f  K3 %.2f  K4 %.2f  K5 %.2f" % (
        e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3), e3.elapsed_time(e4), e4.elapsed_time(e5)))
# unsynchronised bursts, like bench.py's timed region
with torch.no_grad():
    for burst in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(10):
            model(**samples[i % 2])
        t_host = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) * 1e3
        print(f"burst of 10: host enqueue {t_host:.1f} ms, total {t_all:.1f} ms -> {t_all/10:.2f} ms/step", flush=True)
