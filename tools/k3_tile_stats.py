#!/usr/bin/env python3
"""Host-side count for the K3 tile kernel (warp_variance_tile.hip): share of (tile, 8-plane chunk)s whose source footprints
fit the LDS window for every view (pass 0) at a BASELINE config, and the mean footprint size.  CPU only (numpy)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_common as gc  # noqa: E402

CONFIGS = {1: (448, 640, 2, 128), 2: (768, 1152, 4, 256), 3: (896, 1216, 4, 256), 4: (704, 1280, 6, 512)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--tw", type=int, default=8)
    ap.add_argument("--win", type=int, default=128)
    ap.add_argument("--planes", type=int, default=8)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    H, W, V, D = CONFIGS[a.config]
    h, w = H // 4, W // 4
    th = 32 // a.tw
    rng = np.random.default_rng(a.seed)
    rng.standard_normal((V + 1, 1, 32, h, w)).astype(np.float32)  # same stream position as tools/bench_k3.py
    K = gc.synthetic_intrinsics(H, W)
    Ks = K.copy(); Ks[:2] *= 0.25
    def proj(T):
        P = T.copy(); P[:3, :4] = Ks @ P[:3, :4]
        return P
    key_inv = np.linalg.inv(proj(np.eye(4, dtype=np.float32)))
    Ms = [(proj(gc.synthetic_pose(rng)) @ key_inv)[:3] for _ in range(V)]
    depth = np.linspace(0.5, 10.0, D, dtype=np.float32)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    P = a.planes
    nchunk = (D + P - 1) // P
    ty, tx = (h + th - 1) // th, (w + a.tw - 1) // a.tw
    fits_all = np.ones((nchunk, ty, tx), bool)
    area = np.zeros((V, nchunk, ty, tx))
    for v, M in enumerate(Ms):
        for c in range(nchunk):
            cx0 = np.full((ty, tx), 1e9); cx1 = -cx0.copy(); cy0 = cx0.copy(); cy1 = cx1.copy()
            for d in depth[c * P:(c + 1) * P]:
                X = (M[0, 0] * xs + M[0, 1] * ys + M[0, 2]) * d + M[0, 3]
                Y = (M[1, 0] * xs + M[1, 1] * ys + M[1, 2]) * d + M[1, 3]
                Z = (M[2, 0] * xs + M[2, 1] * ys + M[2, 2]) * d + M[2, 3]
                ix = np.clip(X / Z * w / (w - 1) - 0.5, -1, w); iy = np.clip(Y / Z * h / (h - 1) - 0.5, -1, h)
                fx, fy = np.floor(ix), np.floor(iy)
                pad = lambda a_, fill: np.pad(a_, ((0, ty * th - h), (0, tx * a.tw - w)), constant_values=fill)
                r = lambda a_, f, fill: f(f(pad(a_, fill).reshape(ty, th, tx, a.tw), axis=3), axis=1)
                cx0 = np.minimum(cx0, r(fx, np.min, 1e9)); cx1 = np.maximum(cx1, r(fx, np.max, -1e9))
                cy0 = np.minimum(cy0, r(fy, np.min, 1e9)); cy1 = np.maximum(cy1, r(fy, np.max, -1e9))
            ar = (cx1 - cx0 + 2) * (cy1 - cy0 + 2)
            area[v, c] = ar
            fits_all[c] &= ar <= a.win
    print(f"config {a.config}: tile {a.tw}x{th}, {P}-plane chunks, window {a.win} px")
    print(f"  units fitting: {np.mean(area <= a.win):.3f}; chunks with every view fitting (pass 0): {fits_all.mean():.3f}")
    print(f"  mean footprint of fitting units: {area[area <= a.win].mean():.1f} px; per chunk index (share in pass 0):")
    print("  " + " ".join(f"{fits_all[c].mean():.2f}" for c in range(nchunk)))


if __name__ == "__main__":
    main()
