"""The CPU oracle against the golden vectors made from the reference's own CPU path
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

import gen_common as gc
from conftest import load_golden, unpack_mask
from oracle import mvd_oracle as O

ATOL = RTOL = 1e-4  # SURVEY.md 8(c) block-level tolerance, fp32


def test_g1_invdepths():
    g = load_golden("g1_invdepths")
    for S in (64, 256):
        for typ in ("linear_invdepth", "linear_depth"):
            got = O.compute_sampling_invdepths(0.4, 1000.0, S, typ)
            np.testing.assert_allclose(got, g[f"S{S}_{typ}"], rtol=1e-6, atol=1e-9)
    got = O.compute_sampling_invdepths(np.array([0.4, 0.7], np.float32), np.array([1000.0, 50.0], np.float32), 16)
    np.testing.assert_allclose(got, g["batched_S16"], rtol=1e-6, atol=1e-9)


def _sweep_inputs(g, V):
    if "feat_key" in g.files:
        fk = g["feat_key"]
        fs = [g[f"feat_src{v}"] for v in range(V)]
    else:
        shape = tuple(g["shape"])
        fk = gc.rng_array(int(g["seed_key"]), shape)
        fs = [gc.rng_array(int(s), shape) for s in g["seed_src"]]
    Ks = [g[f"K_src{v}"] if f"K_src{v}" in g.files else g["K_key"] for v in range(V)]
    Ts = [g[f"T{v}"] for v in range(V)]
    return fk, fs, g["K_key"], Ks, Ts


def check_sweep_outputs(g, V, corrs, masks, atol=ATOL, rtol=RTOL):
    for v in range(V):
        ref_corr = g[f"corr{v}"]
        ref_mask = unpack_mask(g[f"mask{v}"], ref_corr.shape)
        mism = masks[v] != ref_mask
        # a sample within 1e-4 px of the border may flip its mask; none expected here
        assert mism.mean() <= 1e-4, f"view {v}: {mism.sum()} mask mismatches"
        ok = ~mism
        np.testing.assert_allclose(corrs[v][ok], ref_corr[ok], atol=atol, rtol=rtol)


@pytest.mark.parametrize("name,V", [("g2_sweep_toy", 2), ("g2_sweep_rot", 1), ("g2_sweep_behind", 1),
                                    ("g2_sweep_c256", 2), ("g2_sweep_cfg1", 1)])
def test_g2_sweep(name, V):
    g = load_golden(name)
    fk, fs, Kk, Ks, Ts = _sweep_inputs(g, V)
    inv = g["invdepths"][:, :, 0, 0]
    h, w = fk.shape[-2:]
    if "us0" in g.files:  # grids + visibility (rows A2, A3)
        for v in range(V):
            co = O.epipolar_coeffs(Kk, Ks[v], Ts[v], h, w, fs[v].shape[2], fs[v].shape[3])
            us, vs, vis = O.sweep_grids(co, np.broadcast_to(inv, (fk.shape[0], inv.shape[1])))
            np.testing.assert_allclose(us, g[f"us{v}"], rtol=2e-5, atol=1e-4)
            np.testing.assert_allclose(vs, g[f"vs{v}"], rtol=2e-5, atol=1e-4)
            assert (vis == unpack_mask(g[f"vis{v}"], vis.shape).astype(bool)).all()
    corrs, masks, inv_out = O.planesweep_correlation(fk, Kk, fs, Ts, Ks, sampling_invdepths=inv)
    assert inv_out.shape == g["invdepths"].shape
    check_sweep_outputs(g, V, corrs, masks)


def fusion_weights():
    shapes = {"corr_to_view_weight.0.weight": (128, 256, 3, 3), "corr_to_view_weight.0.bias": (128,),
              "corr_to_view_weight.2.weight": (1, 128, 1, 1), "corr_to_view_weight.2.bias": (1,)}
    return gc.fill_state_dict(shapes, 300)


def fusion_inputs(V):
    rng = np.random.default_rng(310 + V)
    corrs = [rng.standard_normal((2, 256, 12, 18)).astype(np.float32) for _ in range(V)]
    masks = [(rng.uniform(size=(2, 256, 12, 18)) > 0.35).astype(np.float32) for _ in range(V)]
    masks[0][:, :, :3, :4] = 0
    if V > 1:
        for mk in masks:
            mk[:, 5:9, 6:, 9:] = 0
    corrs = [c * mk for c, mk in zip(corrs, masks)]
    return corrs, masks


@pytest.mark.parametrize("V", [1, 2, 4])
def test_g3_fusion(V):
    g = load_golden("g3_fusion")
    sd = fusion_weights()
    corrs, masks = fusion_inputs(V)
    scores = [O.fusion_scores(c, sd["corr_to_view_weight.0.weight"], sd["corr_to_view_weight.0.bias"],
                              sd["corr_to_view_weight.2.weight"], sd["corr_to_view_weight.2.bias"]) for c in corrs]
    fused, fmask = O.fuse_views(corrs, masks, scores)
    ref_fused = g[f"V{V}_fused"]
    assert (fmask == unpack_mask(g[f"V{V}_fmask"], ref_fused.shape)).all()
    np.testing.assert_allclose(fused, ref_fused, atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("name,V", [("g4_warpvar_a", 1), ("g4_warpvar_b", 2), ("g4_warpvar_c", 2)])
def test_g4_warp_variance(name, V):
    g = load_golden(name)
    feats = [g[f"feat{i}"] for i in range(V + 1)]
    projs = [g[f"src_proj{v}"] for v in range(V)]
    if "warped0" in g.files:
        w0 = O.homo_warp(feats[1], projs[0], g["key_proj_inv"], g["depth_values"])
        np.testing.assert_allclose(w0, g["warped0"], atol=ATOL, rtol=RTOL)
    var = O.warp_variance(feats[0], feats[1:], projs, g["key_proj_inv"], g["depth_values"])
    np.testing.assert_allclose(var, g["variance"], atol=ATOL, rtol=RTOL)


def costreg_shapes():
    s = {}
    for name, (ci, co) in dict(conv0=(32, 8), conv1=(8, 16), conv2=(16, 16), conv3=(16, 32), conv4=(32, 32),
                               conv5=(32, 64), conv6=(64, 64)).items():
        s[f"{name}.conv.weight"] = (co, ci, 3, 3, 3)
        for p in ("weight", "bias", "running_mean", "running_var"):
            s[f"{name}.bn.{p}"] = (co,)
    for name, (ci, co) in dict(conv7=(64, 32), conv9=(32, 16), conv11=(16, 8)).items():
        s[f"{name}.0.weight"] = (ci, co, 3, 3, 3)
        for p in ("weight", "bias", "running_mean", "running_var"):
            s[f"{name}.1.{p}"] = (co,)
    s["prob.weight"] = (1, 8, 3, 3, 3)
    s["prob.bias"] = (1,)
    return s


def featurenet_shapes():
    s = {}
    for i, (ci, co, k) in enumerate([(3, 8, 3), (8, 8, 3), (8, 16, 5), (16, 16, 3), (16, 16, 3), (16, 32, 5), (32, 32, 3)]):
        s[f"conv{i}.conv.weight"] = (co, ci, k, k)
        for p in ("weight", "bias", "running_mean", "running_var"):
            s[f"conv{i}.bn.{p}"] = (co,)
    s["feature.weight"] = (32, 32, 3, 3)
    s["feature.bias"] = (32,)
    return s


def test_g9_featurenet():
    """The oracle's FeatureNet (torch-CPU layers, oracle/pipeline.py) against the reference module's outputs."""
    from oracle import pipeline as P
    g = load_golden("g9_featurenet")
    sd = gc.fill_state_dict(featurenet_shapes(), int(g["weight_seed"]))
    x = gc.rng_array(int(g["x_seed"]), (2, 3, 52, 76), 0.5)
    np.testing.assert_allclose(P.feature_net(x, sd, prefix=""), g["out"], atol=ATOL, rtol=RTOL)
    bn = lambda i: tuple(sd[f"conv{i}.bn.{p}"] for p in ("weight", "bias", "running_mean", "running_var"))
    c0 = P.conv_bn_relu_2d(x, sd["conv0.conv.weight"], bn(0))
    np.testing.assert_allclose(c0, g["conv0"], atol=ATOL, rtol=RTOL)
    c2 = P.conv_bn_relu_2d(P.conv_bn_relu_2d(c0, sd["conv1.conv.weight"], bn(1)), sd["conv2.conv.weight"], bn(2), stride=2)
    np.testing.assert_allclose(c2, g["conv2"], atol=ATOL, rtol=RTOL)


def test_g5_costreg():
    g = load_golden("g5_costreg")
    sd = gc.fill_state_dict(costreg_shapes(), int(g["weight_seed"]))
    x = np.abs(gc.rng_array(int(g["x_seed"]), (1, 32, 16, 16, 24), 0.7))
    out, mids = O.cost_reg_net(x, sd, return_all=True)
    np.testing.assert_allclose(mids["conv0"], g["conv0"], atol=ATOL, rtol=RTOL)
    np.testing.assert_allclose(mids["conv1"], g["conv1"], atol=ATOL, rtol=RTOL)
    np.testing.assert_allclose(out, g["out"], atol=2e-4, rtol=1e-3)


@pytest.mark.parametrize("name", ["a", "b"])
def test_g6_regress(name):
    g = load_golden("g6_regress")
    B, D, h, w = g[f"{name}_shape"]
    cost = gc.rng_array(int(g[f"{name}_seed"]), (B, D, h, w), float(g[f"{name}_scale"]))
    dv = np.stack([np.linspace(0.5, 10.0, D, dtype=np.float32)] * B)
    depth, conf, idx = O.softmax_regress(cost, dv)
    np.testing.assert_allclose(depth, g[f"{name}_depth"], atol=1e-5, rtol=1e-5)
    same = idx == g[f"{name}_idx"]
    assert same.mean() > 0.999  # an expected index within float rounding of an integer may truncate differently
    np.testing.assert_allclose(conf[same], g[f"{name}_conf"][same], atol=1e-5, rtol=1e-5)


# ------------------------------------------------------------------------------------------------
# the C/OpenMP part of the oracle (oracle/mvd_oracle_c.c) against the same golden vectors
# ------------------------------------------------------------------------------------------------
from oracle import c_oracle as CO  # noqa: E402


@pytest.mark.parametrize("name,V", [("g2_sweep_toy", 2), ("g2_sweep_rot", 1), ("g2_sweep_behind", 1),
                                    ("g2_sweep_c256", 2), ("g2_sweep_cfg1", 1)])
def test_c_oracle_sweep(name, V):
    g = load_golden(name)
    fk, fs, Kk, Ks, Ts = _sweep_inputs(g, V)
    corrs, masks = CO.sweep_corr(fk, fs, Kk, Ks, Ts, g["invdepths"][:, :, 0, 0])
    check_sweep_outputs(g, V, corrs, masks)


@pytest.mark.parametrize("V", [2, 4])
def test_c_oracle_fusion(V):
    g = load_golden("g3_fusion")
    sd = fusion_weights()
    corrs, masks = fusion_inputs(V)
    scores = [O.fusion_scores(c, sd["corr_to_view_weight.0.weight"], sd["corr_to_view_weight.0.bias"],
                              sd["corr_to_view_weight.2.weight"], sd["corr_to_view_weight.2.bias"]) for c in corrs]
    fused, fmask = CO.fuse_views(corrs, masks, scores)
    ref_fused = g[f"V{V}_fused"]
    assert (fmask == unpack_mask(g[f"V{V}_fmask"], ref_fused.shape)).all()
    np.testing.assert_allclose(fused, ref_fused, atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("name,V", [("g4_warpvar_a", 1), ("g4_warpvar_b", 2), ("g4_warpvar_c", 2)])
def test_c_oracle_warp_variance(name, V):
    g = load_golden(name)
    feats = [g[f"feat{i}"] for i in range(V + 1)]
    projs = [g[f"src_proj{v}"] for v in range(V)]
    if "warped0" in g.files:
        w0 = CO.homo_warp(feats[1], projs[0], g["key_proj_inv"], g["depth_values"])
        np.testing.assert_allclose(w0, g["warped0"], atol=ATOL, rtol=RTOL)
    var = CO.warp_variance(feats[0], feats[1:], projs, g["key_proj_inv"], g["depth_values"])
    np.testing.assert_allclose(var, g["variance"], atol=ATOL, rtol=RTOL)


def test_c_oracle_costreg_and_regress():
    g = load_golden("g5_costreg")
    sd = gc.fill_state_dict(costreg_shapes(), int(g["weight_seed"]))
    x = np.abs(gc.rng_array(int(g["x_seed"]), (1, 32, 16, 16, 24), 0.7))
    out = CO.cost_reg_net(x, sd)
    np.testing.assert_allclose(out, g["out"], atol=2e-4, rtol=1e-3)
    g6 = load_golden("g6_regress")
    B, D, h, w = g6["a_shape"]
    cost = gc.rng_array(int(g6["a_seed"]), (B, D, h, w), float(g6["a_scale"]))
    dv = np.stack([np.linspace(0.5, 10.0, D, dtype=np.float32)] * B)
    depth, conf = CO.softmax_regress(cost, dv)
    np.testing.assert_allclose(depth, g6["a_depth"], atol=1e-5, rtol=1e-5)
    assert np.isclose(conf, g6["a_conf"], atol=1e-5, rtol=1e-5).mean() > 0.999


# ------------------------------------------------------------------------------------------------
# end-to-end oracle pipelines (hot path on the C oracle, adjacent 2-D CNN on torch CPU)
# ------------------------------------------------------------------------------------------------
def test_pipeline_mvsnet_g8():
    import robustmvd_amd as R
    from oracle import pipeline as PL
    g = load_golden("g8_mvsnet")
    H, W, D, V = (int(v) for v in g["shape"])
    shapes = {k: tuple(v.shape) for k, v in R.MVSNet(num_sampling_steps=D).state_dict().items()}
    sd = gc.fill_state_dict(shapes, int(g["weight_seed"]))
    s = gc.synthetic_sample(int(g["sample_seed"]), H, W, V)
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(3, 1, 1)
    images = [((im / 255.0 - mean) / std).astype(np.float32)[None] for im in s["images"]]
    pred = PL.mvsnet_forward(images, [p[None] for p in s["poses"]], [k[None] for k in s["intrinsics"]], 0,
                             (0.5, 10.0), sd, D)
    np.testing.assert_allclose(pred["depth"], g["depth"], rtol=1e-3)
    np.testing.assert_allclose(pred["depth_uncertainty"], g["depth_uncertainty"], atol=2e-3)


def test_pipeline_robustmvd_g7():
    import robustmvd_amd as R
    from oracle import pipeline as PL
    g = load_golden("g7_robustmvd")
    shapes = {k: tuple(v.shape) for k, v in R.RobustMVD().state_dict().items()}
    sd = gc.robustmvd_weights(shapes, int(g["weight_seed"]))
    H, W = 384, 576
    images = [(g[k].astype(np.float32) / 255.0 - 0.4)[None] for k in ("image_key", "image_src0")]
    K_rel = (g["K"] / np.array([[W] * 3, [H] * 3, [1.0] * 3], np.float32))[None]
    pred = PL.robustmvd_forward(images, [np.eye(4, dtype=np.float32)[None], g["T0"][None]], [K_rel, K_rel], 0, sd)
    np.testing.assert_allclose(pred["invdepth"], g["invdepth"][None], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(pred["invdepth_log_b"], g["invdepth_log_b"][None], atol=1e-4, rtol=1e-4)


# ---------------------------------------------------------------------------------------------
# backward restatements against autograd through the reference (g10)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["a", "c"])
def test_g10_warp_variance_backward(name):
    g, g4 = load_golden("g10_grads"), load_golden(f"g4_warpvar_{name}")
    V = len([k for k in g4.files if k.startswith("src_proj")])
    feats = [g4[f"feat{i}"] for i in range(V + 1)]
    G = gc.rng_array(int(g[f"k3_{name}_G_seed"]), g4["variance"].shape)
    dkey, dsrcs = O.warp_variance_backward(feats[0], feats[1:], [g4[f"src_proj{v}"] for v in range(V)], g4["key_proj_inv"],
                                           g4["depth_values"], G)
    np.testing.assert_allclose(dkey, g[f"k3_{name}_dfeat0"], atol=2e-4, rtol=1e-4)
    for v in range(V):
        np.testing.assert_allclose(dsrcs[v], g[f"k3_{name}_dfeat{v + 1}"], atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("name,V", [("toy", 2), ("behind", 1)])
def test_g10_sweep_corr_backward(name, V):
    g = load_golden("g10_grads")
    fk = gc.rng_array(1101, (1, 64, 12, 18))
    fs = [gc.rng_array(1102 + i, (1, 64, 12, 18)) for i in range(V)]
    Ts = [g[f"k1_{name}_T{v}"] for v in range(V)]
    K = g[f"k1_{name}_K"]
    inv = g[f"k1_{name}_invdepths"][:, :, 0, 0]
    corrs, _, _ = O.planesweep_correlation(fk, K, fs, Ts, sampling_invdepths=inv)
    Gs = [gc.rng_array(1110 + v, corrs[v].shape) for v in range(V)]
    for v in range(V):
        np.testing.assert_allclose(corrs[v], g[f"k1_{name}_corr{v}"], atol=ATOL, rtol=RTOL)
    dkey, dsrcs = O.planesweep_correlation_backward(fk, K, fs, Ts, inv, Gs)
    np.testing.assert_allclose(dkey, g[f"k1_{name}_dkey"], atol=2e-4, rtol=1e-4)
    for v in range(V):
        np.testing.assert_allclose(dsrcs[v], g[f"k1_{name}_dsrc{v}"], atol=2e-4, rtol=1e-4)


def test_g10_fusion_backward():
    """K2's VJP w.r.t. the correlation volumes (checked directly) and w.r.t. the score maps (checked through the
    parameter gradients of the score convolutions, which torch back-propagates from dscore)."""
    import torch
    import torch.nn.functional as F
    g = load_golden("g10_grads")
    shapes = {"corr_to_view_weight.0.weight": (128, 256, 3, 3), "corr_to_view_weight.0.bias": (128,),
              "corr_to_view_weight.2.weight": (1, 128, 1, 1), "corr_to_view_weight.2.bias": (1,)}
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in gc.fill_state_dict(shapes, 1200).items()}
    rng = np.random.default_rng(1201)
    V = 3
    masks = [(rng.uniform(size=(1, 256, 10, 14)) > 0.35).astype(np.float32) for _ in range(V)]
    for mk in masks:
        mk[:, 5:9, 6:, 9:] = 0
    corrs = [rng.standard_normal((1, 256, 10, 14)).astype(np.float32) * mk for mk in masks]
    scores_t = []
    for c in corrs:
        hid = F.relu(F.conv2d(torch.from_numpy(c), sd["corr_to_view_weight.0.weight"], sd["corr_to_view_weight.0.bias"], 1, 1))
        scores_t.append(F.conv2d(hid, sd["corr_to_view_weight.2.weight"], sd["corr_to_view_weight.2.bias"]))
    scores = [s.detach().numpy() for s in scores_t]
    fused, _ = O.fuse_views(corrs, masks, scores)
    np.testing.assert_allclose(fused, g["k2_fused"], atol=ATOL, rtol=RTOL)
    G = gc.rng_array(1202, fused.shape)
    dcorrs, dscores = O.fuse_views_backward(corrs, masks, scores, G)
    torch.autograd.backward(scores_t, [torch.from_numpy(d) for d in dscores])
    # the reference's gradient w.r.t. corr_v also contains the path through the score convolutions
    for v in range(V):
        cv = torch.from_numpy(corrs[v]).requires_grad_(True)
        hid = F.relu(F.conv2d(cv, sd["corr_to_view_weight.0.weight"].detach(), sd["corr_to_view_weight.0.bias"].detach(), 1, 1))
        sc = F.conv2d(hid, sd["corr_to_view_weight.2.weight"].detach(), sd["corr_to_view_weight.2.bias"].detach())
        sc.backward(torch.from_numpy(dscores[v]))
        np.testing.assert_allclose(dcorrs[v] + cv.grad.numpy(), g[f"k2_dcorr{v}"], atol=2e-4, rtol=1e-3)
    for k, prm in sd.items():
        np.testing.assert_allclose(prm.grad.numpy(), g["k2_d" + k], atol=5e-4, rtol=1e-3)


# ---------------------------------------------------------------------------------------------
# other sweep consumers (g11): CVP-MVSNet proj_cost and Vis-MVSNet group-wise correlation, made by
# running the reference's (CUDA-only) functions on CPU with Tensor.cuda patched to the identity (make_golden.py::g11)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["pp", "pl"])
def test_g11_cvp_proj_cost(name):
    g = load_golden("g11_sweep_modes")
    got = O.cvp_proj_cost(g["cvp_ref"], [g["cvp_src0"], g["cvp_src1"]], g["cvp_ref_in"], g["cvp_src_in"], g["cvp_ref_ex"],
                          g["cvp_src_ex"], g[f"cvp_hyp_{name}"], alias_bug=True)
    np.testing.assert_allclose(got, g[f"cvp_{name}_cost"], atol=ATOL, rtol=RTOL)
    fixed = O.cvp_proj_cost(g["cvp_ref"], [g["cvp_src0"], g["cvp_src1"]], g["cvp_ref_in"], g["cvp_src_in"], g["cvp_ref_ex"],
                            g["cvp_src_ex"], g[f"cvp_hyp_{name}"], alias_bug=False)
    assert np.abs(fixed - g[f"cvp_{name}_cost"]).max() > 0.1  # the aliasing is not a rounding matter
    assert fixed.min() > -1e-4                                 # ... and the un-aliased form is a true variance


@pytest.mark.parametrize("name", ["s", "p"])
def test_g11_vis_cost_volumes(name):
    g = load_golden("g11_sweep_modes")
    got = O.vis_cost_volumes(g["vis_ref"], g["vis_ref_cam"], [g["vis_src0"], g["vis_src1"]], [g["vis_src_cam0"], g["vis_src_cam1"]],
                             5, g[f"vis_ds_{name}"], g[f"vis_di_{name}"], groups=8)
    for v in range(2):
        np.testing.assert_allclose(got[v], g[f"vis_{name}_cost{v}"], atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("name,norm", [("none", False), ("before", "before"), ("pp", "dim")])
def test_g12_sweep_block_options(name, norm):
    """normalize=False / "before" and per-key-pixel sampling inverse depths (planesweep_corr.py:371-394, 465-487)"""
    g = load_golden("g12_sweep_options")
    fk = gc.rng_array(1401, (1, 64, 12, 18))
    fs = [gc.rng_array(1402 + i, (1, 64, 12, 18)) for i in range(2)]
    kw = dict(sampling_invdepths=g["invdepths_pp"]) if name == "pp" else dict(num_sampling_points=8, min_depth=0.4, max_depth=1000.0)
    corrs, masks, _ = O.planesweep_correlation(fk, g["K"], fs, [g["T0"], g["T1"]], normalize=norm, **kw)
    for v in range(2):
        ref_c = g[f"{name}_corr{v}"]
        ref_m = unpack_mask(g[f"{name}_mask{v}"], ref_c.shape)
        assert (masks[v] != ref_m).mean() <= 1e-3
        ok = masks[v] == ref_m
        np.testing.assert_allclose(corrs[v][ok], ref_c[ok], atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("name,norm", [("none", False), ("before", "before"), ("after", True)])
def test_g13_warp_only_block(name, norm):
    """PlanesweepCorrelation(warp_only=True) = WarpOnlyCorr (planesweep_corr.py:107-140): warped source features (N,S,C,h,w)
    and the sampling mask, for the block's three normalisation modes"""
    g = load_golden("g13_warp_only")
    fs = [gc.rng_array(1502, (1, 16, 12, 18)), gc.rng_array(1503, (1, 16, 12, 18))]
    warped, masks = O.planesweep_warp((12, 18), g["K"], fs, [g["T0"], g["T1"]], g["invdepths"], normalize=norm)
    for v in range(2):
        ref_w = g[f"{name}_warped{v}"]
        ref_m = unpack_mask(g[f"{name}_mask{v}"], (1, 6, 12, 18))
        assert warped[v].shape == ref_w.shape == (1, 6, 16, 12, 18)
        assert (masks[v] != ref_m).mean() <= 1e-3
        ok = np.broadcast_to((masks[v] == ref_m)[:, :, None], ref_w.shape)
        np.testing.assert_allclose(warped[v][ok], ref_w[ok], atol=ATOL, rtol=RTOL)
        assert ref_m.sum() > 0 and (ref_m == 0).sum() > 0  # the fixture exercises both mask values
