#!/usr/bin/env python3
"""Sweeps the launch plan (output-channel tile per workgroup, split of the reduction) of the DispNet layers that the default plan
splits, on the experiments library (MVD_C2_BN / MVD_C2_KSPLIT): us per call incl. the reduce pass.  GPU box only."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "robustmvd_amd", "lib_exp", "libmvd_hip_exp.so")
LAYERS = {"conv3_key": "3 2 0 128 256 1 192 288", "conv_redir": "1 1 0 256 32 1 96 144", "conv3_1": "3 1 0 288 256 1 96 144",
          "conv4": "3 2 0 256 512 1 96 144", "conv4_1": "3 1 0 512 512 1 48 72", "conv5": "3 2 0 512 512 1 48 72",
          "conv5_1": "3 1 0 512 512 1 24 36", "conv6": "3 2 0 512 1024 1 24 36", "conv6_1": "3 1 0 1024 1024 1 12 18",
          "deconv_1": "4 2 1 1024 512 1 12 18", "rfeat1": "3 1 0 1026 512 1 24 36", "deconv_2": "4 2 1 512 256 1 24 36",
          "rfeat2": "3 1 0 770 256 1 48 72", "deconv_3": "4 2 1 256 128 1 48 72", "rfeat3": "3 1 0 386 128 1 96 144"}
names = sys.argv[1:] or list(LAYERS)
for name in names:
    row = []
    for bn in (128, 64):
        for ks in (1, 2, 4, 8, 16, 32):
            env = dict(os.environ, MVD_ALT_LIB=EXP, MVD_C2_BN=str(bn), MVD_C2_KSPLIT=str(ks))
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_conv2d_layer.py")] + LAYERS[name].split() + ["20", "time"],
                               env=env, capture_output=True, text=True)
            us = r.stdout.strip().split(")")[-1].split("us")[0].strip() if r.returncode == 0 and "us per call" in r.stdout else "err"
            row.append(f"bn{bn}/k{ks}: {us}")
    print(f"{name:10s} " + "  ".join(row), flush=True)
