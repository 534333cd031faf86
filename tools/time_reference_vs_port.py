#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference; never runs on the GPU box): times the IMPORTED reference's MVSNet.forward
(CPU, num_gpus=0 path) beside this repository's CPU oracle port (oracle/pipeline.py: C/OpenMP hot path + torch-CPU 2-D
feature net) on the same synthetic inputs and weights, with the protocol of BASELINE.md section 3 (1 burn-in, median of N
timed forwards).  Purpose: show that the port bench.py reports as `cpu_baseline` on the GPU box is not slower than the
reference by construction.  Writes one JSON line; BASELINE.md section 4 records the result.
usage: python tools/time_reference_vs_port.py [--config 1] [--timed 3]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import gen_common as gc  # noqa: E402
from _ref_loader import load_reference  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402
from oracle import pipeline as PL  # noqa: E402

CONFIGS = {1: (448, 640, 2, 128), 2: (768, 1152, 4, 256)}
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=1)
ap.add_argument("--timed", type=int, default=3)
args = ap.parse_args()
H, W, V, D = CONFIGS[args.config]
torch.set_grad_enabled(False)
ref = load_reference()
model = ref.mvsnet.MVSNet(num_sampling_steps=D).eval()
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
sd = gc.fill_state_dict(shapes, 0)
full = model.state_dict()
for k, v in sd.items():
    full[k] = torch.from_numpy(v)
model.load_state_dict(full)
mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(3, 1, 1)
std = np.array([0.229, 0.224, 0.225], np.float32).reshape(3, 1, 1)


def inputs(f):
    s = gc.synthetic_sample(f, H, W, V)
    images = [((im / 255.0 - mean) / std).astype(np.float32)[None] for im in s["images"]]
    return images, [p[None] for p in s["poses"]], [k[None] for k in s["intrinsics"]]


t_ref, t_port, err = [], [], []
for f in range(1 + args.timed):
    images, poses, intr = inputs(f)
    t0 = time.perf_counter()
    pred, _ = model(images=[torch.from_numpy(i) for i in images], poses=[torch.from_numpy(p.copy()) for p in poses],
                    intrinsics=[torch.from_numpy(k) for k in intr], keyview_idx=torch.tensor([0]),
                    depth_range=[torch.tensor([0.5]), torch.tensor([10.0])])
    t1 = time.perf_counter()
    out = PL.mvsnet_forward(images, poses, intr, 0, (0.5, 10.0), sd, D)
    t2 = time.perf_counter()
    if f >= 1:
        t_ref.append(t1 - t0)
        t_port.append(t2 - t1)
    err.append(float(np.abs(pred["depth"].numpy() - out["depth"]).max() / np.abs(out["depth"]).max()))
    print(f"frame {f}: reference {t1 - t0:.2f} s, port {t2 - t1:.2f} s, max rel depth diff {err[-1]:.2e}", file=sys.stderr, flush=True)
res = {"config": f"{H}x{W} V{V} D{D} (BASELINE.json configs[{args.config}])", "cpu_threads_torch": torch.get_num_threads(),
       "cpu_threads_openmp": CO.num_threads(), "host_cpus": os.cpu_count(), "burn_in": 1, "timed": args.timed,
       "reference_median_s": float(np.median(t_ref)), "port_median_s": float(np.median(t_port)),
       "reference_maps_per_s": 1.0 / float(np.median(t_ref)), "port_maps_per_s": 1.0 / float(np.median(t_port)),
       "port_speedup_over_reference": float(np.median(t_ref) / np.median(t_port)), "max_rel_depth_diff": max(err)}
print(json.dumps(res))
