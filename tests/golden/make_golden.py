#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own CPU path.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
Inputs come from gen_common.py (numpy PCG64 seeds) or from the reference's sample_data
(K.npy / to_ref_transform.npy / image.png, which are data files); the fixtures store the seeds or
the small inputs plus the reference's outputs.  No reference source text is stored.

Fixture groups:
  g1_invdepths     compute_sampling_invdepths                       planesweep_corr.py:524-555
  g2_sweep_*       PlanesweepCorrelation (grids, masks, corr)       planesweep_corr.py:396-521
  g3_fusion_*      LearnedFusion                                    learned_fusion.py:24-54
  g4_warpvar_*     homo_warp + variance                             blocks/utils.py:222-268, mvsnet.py:124-136
  g5_costreg       CostRegNet                                       mvsnet_components.py:69-123
  g6_regress       softmax + depth_regression + confidence          mvsnet.py:139-160
  g7_robustmvd     RobustMVD end-to-end on sample_data (384x576)    robust_mvd.py:57-99
  g8_mvsnet        MVSNet end-to-end 64x96, D=32, V=2               mvsnet.py:45-168
  g9_featurenet    FeatureNet (2 images 52x76) + per-layer outputs  mvsnet_components.py:44-66
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_loader import load_reference, REF_ROOT  # noqa: E402
import gen_common as gc  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(8)
ref = load_reference()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1e6:.2f} MB")


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


# ---------------------------------------------------------------------------------------------
def sample_data_calib():
    """K (pixel units, 1280x720) and source_to_key transforms of the reference's sample_data."""
    import os.path as osp

    root = osp.join(REF_ROOT, "sample_data")
    K = np.load(osp.join(root, "key", "K.npy")).astype(np.float32)
    key_to_ref = np.load(osp.join(root, "key", "to_ref_transform.npy")).astype(np.float64)
    ref_to_key = np.linalg.inv(key_to_ref)
    Ts = []
    for i in range(6):
        s2r = np.load(osp.join(root, "source", str(i), "to_ref_transform.npy")).astype(np.float64)
        Ts.append((s2r @ ref_to_key).astype(np.float32))
    return K, np.stack(Ts)


def g1():
    out = {}
    for S in (64, 256):
        for typ in ("linear_invdepth", "linear_depth"):
            v = ref.planesweep_corr.compute_sampling_invdepths(0.4, 1000.0, S, typ)
            out[f"S{S}_{typ}"] = v.numpy()
    v = ref.planesweep_corr.compute_sampling_invdepths(
        np.array([0.4, 0.7], np.float32), np.array([1000.0, 50.0], np.float32), 16
    )
    out["batched_S16"] = v.numpy()
    save("g1_invdepths", **out)


def run_sweep(feat_key, feat_srcs, K_key, K_srcs, Ts, S=None, min_depth=0.4, max_depth=1000.0, invdepths=None):
    blk = ref.planesweep_corr.PlanesweepCorrelation()
    kw = dict(
        feat_key=t(feat_key),
        intrinsics_key=t(K_key),
        feat_sources=[t(f) for f in feat_srcs],
        source_to_key_transforms=[t(T) for T in Ts],
        intrinsics_sources=[t(k) for k in K_srcs] if K_srcs is not None else None,
    )
    if invdepths is None:
        kw.update(num_sampling_points=S, min_depth=min_depth, max_depth=max_depth)
    else:
        kw.update(sampling_invdepths=t(invdepths))
    corrs, masks, inv = blk(**kw)
    us = [sp.us.numpy() for sp in blk.sampling_points]
    vs = [sp.vs.numpy() for sp in blk.sampling_points]
    vis = [sp.mask.numpy() for sp in blk.sampling_points]
    return [c.numpy() for c in corrs], [m.numpy() for m in masks], inv.numpy(), us, vs, vis


def g2():
    K_px, T_sd = sample_data_calib()
    K_rel = (K_px / np.array([[1280.0] * 3, [720.0] * 3, [1.0] * 3], np.float32))[None]

    def pack(prefix, corrs, masks, us=None, vs=None, vis=None):
        d = {}
        for v, (c, m) in enumerate(zip(corrs, masks)):
            d[f"{prefix}corr{v}"] = c
            d[f"{prefix}mask{v}"] = np.packbits(m.astype(np.uint8).ravel())
            if us is not None:
                d[f"{prefix}us{v}"], d[f"{prefix}vs{v}"] = us[v], vs[v]
                d[f"{prefix}vis{v}"] = np.packbits(vis[v].astype(np.uint8).ravel())
        return d

    # (a) toy, sample_data calibration, V=2, inputs stored
    fk = gc.rng_array(101, (1, 16, 12, 18))
    fs = [gc.rng_array(102 + i, (1, 16, 12, 18)) for i in range(2)]
    Ts = [T_sd[0][None], T_sd[1][None]]
    corrs, masks, inv, us, vs, vis = run_sweep(fk, fs, K_rel, [K_rel, K_rel], Ts, S=8)
    save("g2_sweep_toy", feat_key=fk, feat_src0=fs[0], feat_src1=fs[1], K_key=K_rel, K_src0=K_rel, K_src1=K_rel,
         T0=Ts[0], T1=Ts[1], invdepths=inv, **pack("", corrs, masks, us, vs, vis))

    # (b) different source size + different source intrinsics + strong rotation + batch of 2
    rng = np.random.default_rng(7)
    fk = gc.rng_array(111, (2, 16, 12, 18))
    fs = [gc.rng_array(112, (2, 16, 10, 20))]
    Kk = np.stack([K_rel[0], K_rel[0] * np.array([[1.1], [0.9], [1.0]], np.float32)])
    Ks = np.stack([K_rel[0] * np.array([[0.8], [1.2], [1.0]], np.float32), K_rel[0]])
    T = np.stack([gc.synthetic_pose(rng, 0.4, 0.3), gc.synthetic_pose(rng, 0.2, 0.5)])
    corrs, masks, inv, us, vs, vis = run_sweep(fk, fs, Kk, [Ks], [T], S=8, min_depth=0.3, max_depth=20.0)
    save("g2_sweep_rot", feat_key=fk, feat_src0=fs[0], K_key=Kk, K_src0=Ks, T0=T, invdepths=inv,
         **pack("", corrs, masks, us, vs, vis))

    # (c) source camera turned far enough that part of the sweep is behind it (visibility mask,
    #     non-finite replacement) + explicit per-sample inverse depths incl. linear_depth ordering
    fk = gc.rng_array(121, (1, 16, 12, 18))
    fs = [gc.rng_array(122, (1, 16, 12, 18))]
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = gc.rot_xyz(0.1, 0.7, -0.05)
    T[:3, 3] = [0.2, -0.1, -0.6]
    inv_in = ref.planesweep_corr.compute_sampling_invdepths(0.4, 5.0, 8, "linear_depth").numpy()
    corrs, masks, inv, us, vs, vis = run_sweep(fk, fs, K_rel, [K_rel], [T[None]], invdepths=inv_in)
    save("g2_sweep_behind", feat_key=fk, feat_src0=fs[0], K_key=K_rel, K_src0=K_rel, T0=T[None], invdepths=inv,
         **pack("", corrs, masks, us, vs, vis))

    # (d) C=256, 24x36, S=64, V=2 (seeds only)
    fk = gc.rng_array(131, (1, 256, 24, 36))
    fs = [gc.rng_array(132 + i, (1, 256, 24, 36)) for i in range(2)]
    Ts = [T_sd[2][None], T_sd[5][None]]
    corrs, masks, inv, *_ = run_sweep(fk, fs, K_rel, [K_rel, K_rel], Ts, S=64)
    save("g2_sweep_c256", seed_key=131, seed_src=np.array([132, 133]), shape=np.array([1, 256, 24, 36]), K_key=K_rel,
         T0=Ts[0], T1=Ts[1], invdepths=inv, **pack("", corrs, masks))

    # (e) BASELINE config 1 at block level: 384x576 -> 48x72 features, C=256, S=64, key + source/0
    fk = gc.rng_array(141, (1, 256, 48, 72))
    fs = [gc.rng_array(142, (1, 256, 48, 72))]
    corrs, masks, inv, *_ = run_sweep(fk, fs, K_rel, [K_rel], [T_sd[0][None]], S=64)
    save("g2_sweep_cfg1", seed_key=141, seed_src=np.array([142]), shape=np.array([1, 256, 48, 72]), K_key=K_rel,
         T0=T_sd[0][None], invdepths=inv, **pack("", corrs, masks))


def g3():
    shapes = {"corr_to_view_weight.0.weight": (128, 256, 3, 3), "corr_to_view_weight.0.bias": (128,),
              "corr_to_view_weight.2.weight": (1, 128, 1, 1), "corr_to_view_weight.2.bias": (1,)}
    sd = gc.fill_state_dict(shapes, 300)
    m = ref.learned_fusion.LearnedFusion().eval()
    m.load_state_dict({k: t(v) for k, v in sd.items()})
    out = {}
    for V in (1, 2, 4):
        rng = np.random.default_rng(310 + V)
        corrs = [rng.standard_normal((2, 256, 12, 18)).astype(np.float32) for _ in range(V)]
        masks = [(rng.uniform(size=(2, 256, 12, 18)) > 0.35).astype(np.float32) for _ in range(V)]
        masks[0][:, :, :3, :4] = 0  # a region where some views are masked
        if V > 1:
            for mk in masks:
                mk[:, 5:9, 6:, 9:] = 0  # a region masked in ALL views -> fused mask 0
        corrs = [c * mk for c, mk in zip(corrs, masks)]
        fused, fmask = m([t(c) for c in corrs], [t(mk) for mk in masks])
        out[f"V{V}_fused"] = fused.numpy()
        out[f"V{V}_fmask"] = np.packbits(fmask.numpy().astype(np.uint8).ravel())
    save("g3_fusion", weight_seed=300, **out)


def mvs_proj(K, T, key):
    """Projection matrices exactly as MVSNet.forward builds them (mvsnet.py:76-91)."""
    Ks = K.copy()
    Ks[:2] *= 0.25
    P = T.copy()
    P[:3, :4] = Ks @ P[:3, :4]
    return np.linalg.inv(P).astype(np.float32) if key else P.astype(np.float32)


def g4():
    for name, (B, D, V, seed, rot, dmin, dmax) in {
        "a": (1, 8, 1, 400, 0.05, 0.5, 10.0),
        "b": (2, 32, 2, 410, 0.05, 0.5, 10.0),
        "c": (1, 8, 2, 420, 0.6, 0.2, 3.0),  # wide baseline: large parts sample outside / behind the source
    }.items():
        h, w, C = 16, 24, 32
        rng = np.random.default_rng(seed)
        K = gc.synthetic_intrinsics(h * 4, w * 4)
        feats = [rng.standard_normal((B, C, h, w)).astype(np.float32) for _ in range(V + 1)]
        depth = np.stack([np.linspace(dmin, dmax, D, dtype=np.float32)] * B)
        key_inv = np.stack([mvs_proj(K, np.eye(4, dtype=np.float32), True)] * B)
        srcP = [np.stack([mvs_proj(K, gc.synthetic_pose(rng, rot, 0.15 if rot < 0.1 else 0.6), False) for _ in range(B)])
                for _ in range(V)]
        vol_sum = t(feats[0]).unsqueeze(2).repeat(1, 1, D, 1, 1)
        vol_sq = vol_sum ** 2
        warped0 = None
        for v in range(V):
            wv = ref.blocks_utils.homo_warp(t(feats[v + 1]), t(srcP[v]), t(key_inv), t(depth))
            if v == 0:
                warped0 = wv.numpy().copy()
            vol_sum = vol_sum + wv
            vol_sq = vol_sq + wv ** 2
        var = vol_sq.div_(V + 1).sub_(vol_sum.div_(V + 1).pow_(2)).numpy()
        d = {f"feat{i}": f for i, f in enumerate(feats)}
        d.update({f"src_proj{v}": p for v, p in enumerate(srcP)})
        if name == "a":
            d["warped0"] = warped0
        save(f"g4_warpvar_{name}", key_proj_inv=key_inv, depth_values=depth, variance=var, **d)


COSTREG_SHAPES = None


def costreg_shapes():
    m = ref.mvsnet_components.CostRegNet()
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def g5():
    sd = gc.fill_state_dict(costreg_shapes(), 500)
    m = ref.mvsnet_components.CostRegNet().eval()
    full = {k: t(v) for k, v in sd.items()}
    for k, v in m.state_dict().items():
        if k.endswith("num_batches_tracked"):
            full[k] = v
    m.load_state_dict(full)
    x = np.abs(gc.rng_array(501, (1, 32, 16, 16, 24), 0.7))
    y = m(t(x)).numpy()
    # per-layer outputs of the first two layers, to localise a mismatch
    c0 = m.conv0(t(x))
    c1 = m.conv1(c0)
    save("g5_costreg", weight_seed=500, x_seed=501, out=y, conv0=c0.numpy(), conv1=c1.numpy())


def featurenet_shapes():
    m = ref.mvsnet_components.FeatureNet()
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def g9():
    sd = gc.fill_state_dict(featurenet_shapes(), 900)
    m = ref.mvsnet_components.FeatureNet().eval()
    full = {k: t(v) for k, v in sd.items()}
    for k, v in m.state_dict().items():
        if k.endswith("num_batches_tracked"):
            full[k] = v
    m.load_state_dict(full)
    x = gc.rng_array(901, (2, 3, 52, 76), 0.5)  # 52x76: ragged against every tile shape, odd sizes after /2 and /4
    y = m(t(x))
    c0 = m.conv0(t(x))
    c2 = m.conv2(m.conv1(c0))
    save("g9_featurenet", weight_seed=900, x_seed=901, out=y.numpy(), conv0=c0.numpy(), conv2=c2.numpy())


def g6():
    out = {}
    for name, (B, D, h, w, seed, scale) in {"a": (2, 32, 16, 24, 600, 3.0), "b": (1, 8, 5, 7, 601, 8.0)}.items():
        cost = gc.rng_array(seed, (B, D, h, w), scale)
        depth_values = np.stack([np.linspace(0.5, 10.0, D, dtype=np.float32)] * B)
        p = torch.softmax(t(cost), 1)
        depth = ref.blocks_utils.depth_regression(p, t(depth_values))
        sum4 = 4 * torch.nn.functional.avg_pool3d(
            torch.nn.functional.pad(p.unsqueeze(1), pad=(0, 0, 0, 0, 1, 2)), (4, 1, 1), stride=1).squeeze(1)
        idx = ref.blocks_utils.depth_regression(p, torch.arange(D, dtype=p.dtype)).long()
        conf = torch.gather(sum4, 1, idx.unsqueeze(1)).squeeze(1)
        out.update({f"{name}_seed": seed, f"{name}_scale": scale, f"{name}_shape": np.array([B, D, h, w]),
                    f"{name}_depth": depth.numpy(), f"{name}_conf": conf.numpy(), f"{name}_idx": idx.numpy()})
    save("g6_regress", **out)


def g7():
    from PIL import Image
    import os.path as osp

    H, W = 384, 576
    root = osp.join(REF_ROOT, "sample_data")
    K_px, T_sd = sample_data_calib()
    imgs = []
    for p in (osp.join(root, "key", "image.png"), osp.join(root, "source", "0", "image.png")):
        im = Image.open(p).convert("RGB").resize((W, H), Image.BILINEAR)
        imgs.append(np.asarray(im, dtype=np.uint8).transpose(2, 0, 1))
    K = K_px * np.array([[W / 1280.0] * 3, [H / 720.0] * 3, [1.0] * 3], np.float32)
    model = ref.robust_mvd.RobustMVD().eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = gc.robustmvd_weights(shapes, 700)
    model.load_state_dict({k: t(v) for k, v in sd.items()})
    ref.helpers.add_run_function(model)
    sample = dict(images=[imgs[0].astype(np.float32), imgs[1].astype(np.float32)],
                  intrinsics=[K.copy(), K.copy()],
                  poses=[np.eye(4, dtype=np.float32), T_sd[0]], keyview_idx=0)
    pred, aux = model.run(**sample)
    # intermediate: fused correlation volume (input of the 2-D cost-volume encoder)
    b_images, b_key, b_poses, b_intr, _ = ref.helpers.add_batch_dim(
        sample["images"], 0, sample["poses"], sample["intrinsics"])
    inp = model.input_adapter(images=b_images, keyview_idx=b_key, poses=b_poses, intrinsics=b_intr)
    enc_key = model.encoder(inp["images"][0])[1]
    enc_src = model.encoder(inp["images"][1])[1]
    corrs, masks, _ = model.corr_block(feat_key=enc_key, intrinsics_key=inp["intrinsics"][0], feat_sources=[enc_src],
                                       source_to_key_transforms=[inp["poses"][1]], intrinsics_sources=[inp["intrinsics"][1]],
                                       num_sampling_points=256, min_depth=0.4, max_depth=1000.0)
    save("g7_robustmvd", image_key=imgs[0], image_src0=imgs[1], K=K, T0=T_sd[0], weight_seed=700,
         depth=pred["depth"], depth_uncertainty=pred["depth_uncertainty"],
         invdepth=aux["invdepth"], invdepth_log_b=aux["invdepth_log_b"],
         invdepths_all_0=aux["invdepths_all"][0], invdepths_all_3=aux["invdepths_all"][3],
         corr0_sub=corrs[0].numpy()[:, ::4, ::2, ::2], mask0=np.packbits(masks[0].numpy().astype(np.uint8).ravel()),
         enc_key_sub=enc_key.numpy()[:, ::16, ::4, ::4])

    # the same weights with 2 source views at 128x192 exercise the learned fusion inside the model
    H2, W2 = 128, 192
    rng = np.random.default_rng(710)
    images = [rng.uniform(0, 255, (3, H2, W2)).astype(np.float32) for _ in range(3)]
    K2 = gc.synthetic_intrinsics(H2, W2)
    poses = [gc.synthetic_pose(rng), np.eye(4, dtype=np.float32), gc.synthetic_pose(rng)]
    pred, aux = model.run(images=images, intrinsics=[K2, K2, K2], poses=poses, keyview_idx=1)
    save("g7_robustmvd_v2", seed=710, weight_seed=700, depth=pred["depth"], invdepth=aux["invdepth"],
         invdepth_log_b=aux["invdepth_log_b"], depth_uncertainty=pred["depth_uncertainty"])


def g8():
    H, W, D, V = 64, 96, 32, 2
    model = ref.mvsnet.MVSNet(num_sampling_steps=D).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = gc.fill_state_dict(shapes, 800)
    full = {k: t(v) for k, v in sd.items()}
    for k, v in model.state_dict().items():
        if k.endswith("num_batches_tracked"):
            full[k] = v
    model.load_state_dict(full)
    s = gc.synthetic_sample(801, H, W, V)
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(3, 1, 1)
    images = [t(((im / 255.0 - mean) / std).astype(np.float32)[None]) for im in s["images"]]
    poses = [t(p[None].copy()) for p in s["poses"]]
    intr = [t(k[None].copy()) for k in s["intrinsics"]]
    pred, _ = model(images=images, poses=poses, intrinsics=intr, keyview_idx=torch.tensor([0]),
                    depth_range=[torch.tensor([0.5]), torch.tensor([10.0])])
    save("g8_mvsnet", sample_seed=801, weight_seed=800, shape=np.array([H, W, D, V]),
         depth=pred["depth"].numpy(), depth_uncertainty=pred["depth_uncertainty"].numpy())


def g10():
    """Gradients by autograd THROUGH THE REFERENCE (backward of K1 / K3 w.r.t. the feature maps,
    grids under no_grad; multi_view_depth_training.py:231-246 back-propagates through exactly these ops), plus
    LearnedFusion's.  loss = sum(output * G) with a seeded random G, so the stored gradients are vector-Jacobian products."""
    out = {}
    with torch.enable_grad():
        # --- K3: homo_warp + variance (blocks/utils.py:222-268, mvsnet.py:124-135) on the inputs of g4_warpvar_a / _c
        for name in ("a", "c"):
            g = np.load(os.path.join(HERE, f"g4_warpvar_{name}.npz"))
            V = len([k for k in g.files if k.startswith("src_proj")])
            feats = [t(g[f"feat{i}"]).requires_grad_(True) for i in range(V + 1)]
            depth, key_inv = t(g["depth_values"]), t(g["key_proj_inv"])
            D = depth.shape[1]
            vol_sum = feats[0].unsqueeze(2).repeat(1, 1, D, 1, 1)
            vol_sq = vol_sum ** 2
            for v in range(V):
                wv = ref.blocks_utils.homo_warp(feats[v + 1], t(g[f"src_proj{v}"]), key_inv, depth)
                vol_sum = vol_sum + wv
                vol_sq = vol_sq + wv ** 2
            var = vol_sq / (V + 1) - (vol_sum / (V + 1)) ** 2
            np.testing.assert_allclose(var.detach().numpy(), g["variance"], atol=1e-5, rtol=1e-5)
            G = gc.rng_array(1000 + ord(name), tuple(var.shape))
            (var * t(G)).sum().backward()
            out[f"k3_{name}_G_seed"] = np.int64(1000 + ord(name))
            for i, f in enumerate(feats):
                out[f"k3_{name}_dfeat{i}"] = f.grad.numpy()
        # --- K1: PlanesweepCorrelation (planesweep_corr.py:396-521), C = 64, 12x18, S = 8
        K_px, T_sd = sample_data_calib()
        K_rel = (K_px / np.array([[1280.0] * 3, [720.0] * 3, [1.0] * 3], np.float32))[None]
        Tb = np.eye(4, dtype=np.float32)
        Tb[:3, :3] = gc.rot_xyz(0.1, 0.7, -0.05)
        Tb[:3, 3] = [0.2, -0.1, -0.6]
        for name, Ts, kw in (("toy", [T_sd[0][None], T_sd[1][None]], dict(num_sampling_points=8, min_depth=0.4, max_depth=1000.0)),
                             ("behind", [Tb[None]], dict(num_sampling_points=8, min_depth=0.4, max_depth=5.0,
                                                         sampling_type="linear_depth"))):
            V = len(Ts)
            fk = t(gc.rng_array(1101, (1, 64, 12, 18))).requires_grad_(True)
            fs = [t(gc.rng_array(1102 + i, (1, 64, 12, 18))).requires_grad_(True) for i in range(V)]
            blk = ref.planesweep_corr.PlanesweepCorrelation()
            corrs, masks, inv = blk(feat_key=fk, intrinsics_key=t(K_rel), feat_sources=fs,
                                    source_to_key_transforms=[t(T) for T in Ts], **kw)
            loss = sum((c * t(gc.rng_array(1110 + v, tuple(c.shape)))).sum() for v, c in enumerate(corrs))
            loss.backward()
            out[f"k1_{name}_K"] = K_rel
            out[f"k1_{name}_invdepths"] = inv.detach().numpy()
            for v in range(V):
                out[f"k1_{name}_T{v}"] = Ts[v]
                out[f"k1_{name}_corr{v}"] = corrs[v].detach().numpy()
                out[f"k1_{name}_dsrc{v}"] = fs[v].grad.numpy()
            out[f"k1_{name}_dkey"] = fk.grad.numpy()
        # --- LearnedFusion (learned_fusion.py:24-54), V = 3: gradients w.r.t. the correlation volumes and the parameters
        shapes = {"corr_to_view_weight.0.weight": (128, 256, 3, 3), "corr_to_view_weight.0.bias": (128,),
                  "corr_to_view_weight.2.weight": (1, 128, 1, 1), "corr_to_view_weight.2.bias": (1,)}
        sd = gc.fill_state_dict(shapes, 1200)
        m = ref.learned_fusion.LearnedFusion().eval()
        m.load_state_dict({k: t(v) for k, v in sd.items()})
        rng = np.random.default_rng(1201)
        V = 3
        masks = [(rng.uniform(size=(1, 256, 10, 14)) > 0.35).astype(np.float32) for _ in range(V)]
        for mk in masks:
            mk[:, 5:9, 6:, 9:] = 0
        corrs = [t(rng.standard_normal((1, 256, 10, 14)).astype(np.float32) * mk).requires_grad_(True) for mk in masks]
        fused, fmask = m(corrs, [t(mk) for mk in masks])
        (fused * t(gc.rng_array(1202, tuple(fused.shape)))).sum().backward()
        out["k2_fused"] = fused.detach().numpy()
        for v in range(V):
            out[f"k2_mask{v}"] = np.packbits(masks[v].astype(np.uint8).ravel())
            out[f"k2_dcorr{v}"] = corrs[v].grad.numpy()
        for k, prm in m.named_parameters():
            out["k2_d" + k] = prm.grad.numpy()
    save("g10_grads", **out)


def g11():
    """Other sweep consumers.  The reference's CVP-MVSNet / Vis-MVSNet sweep code is CUDA-only only
    through hard `.cuda()` calls on freshly made constants; with Tensor.cuda patched to the identity the very same functions
    run on CPU: proj_cost (cvp_mvsnet_components.py:375-456) and get_homographies + homography_warping +
    groupwise_correlation as SingleStage.build_cost_volume chains them (vis_mvsnet_singlestage.py:86-122,242)."""
    import types
    from _ref_loader import _load
    torch.Tensor.cuda = lambda self, *a, **k: self
    cvp = _load("rmvd.models.blocks.cvp_mvsnet_components", "rmvd/models/blocks/cvp_mvsnet_components.py")
    bu = ref.blocks_utils
    out = {}
    # ---- cvp: B=1, C=16, 12x18, D=6, 2 sources; per-pixel hypotheses and per-plane hypotheses
    rng = np.random.default_rng(1300)
    B, C, h, w, D, V = 1, 16, 12, 18, 6, 2
    K = gc.synthetic_intrinsics(h * 4, w * 4)
    K[:2] *= 0.25
    ref_f = rng.standard_normal((B, C, h, w)).astype(np.float32)
    src_f = [rng.standard_normal((B, C, h, w)).astype(np.float32) for _ in range(V)]
    ref_ex = np.eye(4, dtype=np.float32)[None]
    src_ex = np.stack([gc.synthetic_pose(rng, 0.08, 0.2) for _ in range(V)])[None]          # (B,V,4,4)
    ref_in = K[None]
    src_in = np.stack([K * np.array([[1.05], [0.97], [1.0]], np.float32), K])[None]          # (B,V,3,3)
    planes = np.linspace(0.6, 6.0, D, dtype=np.float32)
    hyp_pp = (planes[None, :, None, None] * (1.0 + 0.15 * rng.uniform(-1, 1, (B, D, h, w)))).astype(np.float32)
    hyp_pl = np.broadcast_to(planes[None, :, None, None], (B, D, h, w)).astype(np.float32).copy()
    settings = types.SimpleNamespace(nsrc=V, mode="test")
    for name, hyp in (("pp", hyp_pp), ("pl", hyp_pl)):
        cv = cvp.proj_cost(settings, t(ref_f.copy()), [[t(f)] for f in src_f], 0, t(ref_in), t(src_in), t(ref_ex), t(src_ex), t(hyp))
        out[f"cvp_{name}_cost"] = cv.numpy()
    out.update(cvp_ref=ref_f, cvp_src0=src_f[0], cvp_src1=src_f[1], cvp_ref_in=ref_in, cvp_src_in=src_in, cvp_ref_ex=ref_ex,
               cvp_src_ex=src_ex, cvp_hyp_pp=hyp_pp, cvp_hyp_pl=hyp_pl)
    # ---- vis: B=2, C=32 (8 groups), 12x20, D=5, 2 sources; scalar and per-pixel depth_start / depth_interval
    rng = np.random.default_rng(1310)
    B, C, h, w, D, V = 2, 32, 12, 20, 5, 2
    Kv = gc.synthetic_intrinsics(h, w)

    def cam(T):
        c = np.zeros((2, 4, 4), np.float32)
        c[0] = T
        c[1, :3, :3] = Kv
        c[1, 3, 3] = 1
        return c

    ref_cam = np.stack([cam(np.eye(4, dtype=np.float32)), cam(gc.synthetic_pose(rng, 0.02, 0.05))])
    srcs_cam = [np.stack([cam(gc.synthetic_pose(rng, 0.08, 0.2)) for _ in range(B)]) for _ in range(V)]
    ref_f = rng.standard_normal((B, C, h, w)).astype(np.float32)
    srcs_f = [rng.standard_normal((B, C, h, w)).astype(np.float32) for _ in range(V)]
    ds_s = np.full((B, 1, 1, 1), 0.8, np.float32)
    di_s = np.full((B, 1, 1, 1), 0.9, np.float32)
    ds_p = (0.8 + 0.3 * rng.uniform(0, 1, (B, 1, h, w))).astype(np.float32)
    di_p = (0.9 + 0.2 * rng.uniform(0, 1, (B, 1, h, w))).astype(np.float32)
    for name, ds, di in (("s", ds_s, di_s), ("p", ds_p, di_p)):
        for v in range(V):
            Hs = bu.get_homographies(t(ref_cam), t(srcs_cam[v]), D, t(ds), t(di))
            src_nd = t(srcs_f[v]).unsqueeze(1).repeat(1, D, 1, 1, 1).view(-1, C, h, w)
            Hs_flat = Hs.view(-1, *Hs.size()[2:])
            warped = bu.homography_warping(src_nd, Hs_flat).view(-1, D, C, h, w).transpose(1, 2)
            ref_ncdhw = t(ref_f).unsqueeze(2).expand(-1, -1, D, -1, -1)
            out[f"vis_{name}_cost{v}"] = bu.groupwise_correlation(ref_ncdhw, warped, 8, 1).numpy()
    out.update(vis_ref=ref_f, vis_src0=srcs_f[0], vis_src1=srcs_f[1], vis_ref_cam=ref_cam, vis_src_cam0=srcs_cam[0],
               vis_src_cam1=srcs_cam[1], vis_ds_s=ds_s, vis_di_s=di_s, vis_ds_p=ds_p, vis_di_p=di_p)
    save("g11_sweep_modes", **out)


def g12():
    """The other options of the PlanesweepCorrelation block (planesweep_corr.py:371-394, 465-487): normalize=False and
    normalize="before", and sampling inverse depths per key pixel (N,S,H,W).  C = 64, 12x18, S = 8, 2 sources."""
    K_px, T_sd = sample_data_calib()
    K_rel = (K_px / np.array([[1280.0] * 3, [720.0] * 3, [1.0] * 3], np.float32))[None]
    fk = gc.rng_array(1401, (1, 64, 12, 18))
    fs = [gc.rng_array(1402 + i, (1, 64, 12, 18)) for i in range(2)]
    Ts = [T_sd[0][None], T_sd[3][None]]
    out = {"K": K_rel, "T0": Ts[0], "T1": Ts[1]}
    rng = np.random.default_rng(1410)
    base = ref.planesweep_corr.compute_sampling_invdepths(0.4, 1000.0, 8).numpy()            # (1,8)
    inv_pp = (base[:, :, None, None] * (1.0 + 0.2 * rng.uniform(-1, 1, (1, 8, 12, 18)))).astype(np.float32)
    out["invdepths_pp"] = inv_pp
    for name, norm, inv in (("none", False, None), ("before", "before", None), ("pp", "dim", inv_pp)):
        blk = ref.planesweep_corr.PlanesweepCorrelation(normalize=norm)
        kw = dict(feat_key=t(fk), intrinsics_key=t(K_rel), feat_sources=[t(f) for f in fs], source_to_key_transforms=[t(T) for T in Ts])
        if inv is None:
            kw.update(num_sampling_points=8, min_depth=0.4, max_depth=1000.0)
        else:
            kw.update(sampling_invdepths=t(inv))
        corrs, masks, invd = blk(**kw)
        for v in range(2):
            out[f"{name}_corr{v}"] = corrs[v].numpy()
            out[f"{name}_mask{v}"] = np.packbits(masks[v].numpy().astype(np.uint8).ravel())
    save("g12_sweep_options", **out)


def g13():
    """PlanesweepCorrelation(warp_only=True) (WarpOnlyCorr, planesweep_corr.py:107-140): warped source features and sampling
    masks for normalize in (False, "before", True).  C = 16, key and sources 12x18 (warp_multi reshapes the result to the SOURCE size, :33-43, so the reference itself only runs with equal sizes), S = 6."""
    K_px, T_sd = sample_data_calib()
    K_rel = (K_px / np.array([[1280.0] * 3, [720.0] * 3, [1.0] * 3], np.float32))[None]
    fk = gc.rng_array(1501, (1, 16, 12, 18))
    fs = [gc.rng_array(1502, (1, 16, 12, 18)), gc.rng_array(1503, (1, 16, 12, 18))]
    Ts = [T_sd[0][None], T_sd[3][None]]
    out = {"K": K_rel, "T0": Ts[0], "T1": Ts[1]}
    for name, norm in (("none", False), ("before", "before"), ("after", True)):
        blk = ref.planesweep_corr.PlanesweepCorrelation(warp_only=True, normalize=norm)
        warped, masks, invd = blk(feat_key=t(fk), intrinsics_key=t(K_rel), feat_sources=[t(f) for f in fs],
                                  source_to_key_transforms=[t(T) for T in Ts], num_sampling_points=6, min_depth=0.4, max_depth=1000.0)
        for v in range(2):
            out[f"{name}_warped{v}"] = warped[v].numpy()
            out[f"{name}_mask{v}"] = np.packbits(masks[v].numpy().astype(np.uint8).ravel())
    out["invdepths"] = invd.numpy()
    save("g13_warp_only", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13"]
    for g in which:
        globals()[g]()
