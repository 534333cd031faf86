#!/usr/bin/env python3
"""Path A end-to-end difference to the reference's golden output (g7, sample_data pair at 384x576) under the vendor
library's default solver choice and with its Winograd solvers disabled (run once per setting: the environment variable is
read when MIOpen initialises).  Shows where the 2e-3 end-to-end tolerance of Path A comes from.
usage: [MIOPEN_DEBUG_CONV_WINOGRAD=0] python tools/path_a_parity.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen_common as gc
import robustmvd_amd as R
g = np.load(os.path.join(ROOT, "tests", "golden", "g7_robustmvd.npz"))
dev = torch.device("cuda:0")
model = R.RobustMVD().eval()
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in gc.robustmvd_weights(shapes, int(g["weight_seed"])).items()})
model = R.add_run_function(model.to(dev))
pred, aux = model.run(images=[g["image_key"].astype(np.float32), g["image_src0"].astype(np.float32)], intrinsics=[g["K"].copy(), g["K"].copy()],
                      poses=[np.eye(4, dtype=np.float32), g["T0"]], keyview_idx=0)
d = np.abs(aux["invdepth"] - g["invdepth"])
print(f"MIOPEN_DEBUG_CONV_WINOGRAD={os.environ.get('MIOPEN_DEBUG_CONV_WINOGRAD', '(default)')}: "
      f"invdepth max abs diff {d.max():.3e}, mean {d.mean():.3e}, 99.9th pct {np.percentile(d, 99.9):.3e}; invdepth range [{g['invdepth'].min():.3f}, {g['invdepth'].max():.3f}]")
