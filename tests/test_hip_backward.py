"""GPU tests of the backward (VJP) kernels of the sweep operators against
gradients obtained by autograd THROUGH THE REFERENCE (tests/golden/g10_grads.npz, made by make_golden.py::g10) and against
the oracle's restatements on other shapes.  Scatter-adds use float atomics: tolerance atol 2e-4 / rtol 1e-4 like the CPU pin."""
import numpy as np
import pytest
import torch

import gen_common as gc
from conftest import load_golden, unpack_mask
from oracle import mvd_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def T(x, dev, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    return t.requires_grad_(True) if grad else t


@pytest.mark.parametrize("name", ["a", "c"])
def test_warp_variance_backward_golden(name, dev):
    from robustmvd_amd import ops
    g, g4 = load_golden("g10_grads"), load_golden(f"g4_warpvar_{name}")
    V = len([k for k in g4.files if k.startswith("src_proj")])
    feats = [T(g4[f"feat{i}"], dev, grad=True) for i in range(V + 1)]
    var = ops.warp_variance_autograd(feats[0], feats[1:], [T(g4[f"src_proj{v}"], dev) for v in range(V)],
                                     T(g4["key_proj_inv"], dev), T(g4["depth_values"], dev))
    np.testing.assert_allclose(var.detach().cpu().numpy(), g4["variance"], atol=1e-4, rtol=1e-4)
    G = gc.rng_array(int(g[f"k3_{name}_G_seed"]), g4["variance"].shape)
    (var * T(G, dev)).sum().backward()
    for i, f in enumerate(feats):
        np.testing.assert_allclose(f.grad.cpu().numpy(), g[f"k3_{name}_dfeat{i}"], atol=2e-4, rtol=1e-4)


def test_warp_variance_backward_ragged_vs_oracle(dev):
    from robustmvd_amd import ops
    from test_hip_shapes import mvs_inputs
    B, C, h, w, D, V = 2, 32, 13, 21, 5, 3
    feats, projs, key_inv, depth = mvs_inputs(B, C, h, w, D, V, seed=5, rot=0.2, trans=0.3)
    G = gc.rng_array(77, (B, C, D, h, w))
    dkey, dsrcs = O.warp_variance_backward(feats[0], feats[1:], projs, key_inv, depth, G)
    ft = [T(f, dev, grad=True) for f in feats]
    var = ops.warp_variance_autograd(ft[0], ft[1:], [T(p, dev) for p in projs], T(key_inv, dev), T(depth, dev))
    (var * T(G, dev)).sum().backward()
    np.testing.assert_allclose(ft[0].grad.cpu().numpy(), dkey, atol=3e-4, rtol=1e-3)
    for v in range(V):
        np.testing.assert_allclose(ft[v + 1].grad.cpu().numpy(), dsrcs[v], atol=3e-4, rtol=1e-3)


@pytest.mark.parametrize("name,V", [("toy", 2), ("behind", 1)])
def test_sweep_corr_backward_golden(name, V, dev):
    import robustmvd_amd as R
    g = load_golden("g10_grads")
    fk = T(gc.rng_array(1101, (1, 64, 12, 18)), dev, grad=True)
    fs = [T(gc.rng_array(1102 + i, (1, 64, 12, 18)), dev, grad=True) for i in range(V)]
    blk = R.PlanesweepCorrelation()
    corrs, masks, _ = blk(fk, T(g[f"k1_{name}_K"], dev), fs, [T(g[f"k1_{name}_T{v}"], dev) for v in range(V)],
                          sampling_invdepths=T(g[f"k1_{name}_invdepths"], dev))
    assert all(c.requires_grad for c in corrs) and not any(m.requires_grad for m in masks)
    loss = 0
    for v in range(V):
        np.testing.assert_allclose(corrs[v].detach().cpu().numpy(), g[f"k1_{name}_corr{v}"], atol=1e-4, rtol=1e-4)
        loss = loss + (corrs[v] * T(gc.rng_array(1110 + v, tuple(corrs[v].shape)), dev)).sum()
    loss.backward()
    np.testing.assert_allclose(fk.grad.cpu().numpy(), g[f"k1_{name}_dkey"], atol=2e-4, rtol=1e-4)
    for v in range(V):
        np.testing.assert_allclose(fs[v].grad.cpu().numpy(), g[f"k1_{name}_dsrc{v}"], atol=2e-4, rtol=1e-4)


def test_learned_fusion_backward_golden(dev):
    """LearnedFusion end to end (score convolutions on torch, view weighting on K2 forward + backward kernels): gradients
    w.r.t. the correlation volumes and w.r.t. the module's parameters against autograd through the reference module."""
    import robustmvd_amd as R
    g = load_golden("g10_grads")
    shapes = {"corr_to_view_weight.0.weight": (128, 256, 3, 3), "corr_to_view_weight.0.bias": (128,),
              "corr_to_view_weight.2.weight": (1, 128, 1, 1), "corr_to_view_weight.2.bias": (1,)}
    m = R.LearnedFusion()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in gc.fill_state_dict(shapes, 1200).items()})
    m = m.to(dev)
    rng = np.random.default_rng(1201)
    V = 3
    masks = [(rng.uniform(size=(1, 256, 10, 14)) > 0.35).astype(np.float32) for _ in range(V)]
    for mk in masks:
        mk[:, 5:9, 6:, 9:] = 0
    corrs = [T(rng.standard_normal((1, 256, 10, 14)).astype(np.float32) * mk, dev, grad=True) for mk in masks]
    fused, fmask = m(corrs, [T(mk, dev) for mk in masks])
    np.testing.assert_allclose(fused.detach().cpu().numpy(), g["k2_fused"], atol=1e-4, rtol=1e-4)
    (fused * T(gc.rng_array(1202, tuple(fused.shape)), dev)).sum().backward()
    for v in range(V):
        np.testing.assert_allclose(corrs[v].grad.cpu().numpy(), g[f"k2_dcorr{v}"], atol=3e-4, rtol=2e-3)
    for k, prm in m.named_parameters():
        np.testing.assert_allclose(prm.grad.cpu().numpy(), g["k2_d" + k], atol=1e-3, rtol=2e-3)


def test_robustmvd_training_step_runs(dev):
    """create_model("robust_mvd", train=True): one forward + backward through encoder -> K1 -> fusion/K2 -> decoder;
    every parameter that the reference trains receives a finite gradient (the encoder's only through the sweep)."""
    import robustmvd_amd as R
    assert "robust_mvd" in R.list_models(trainable_only=True)
    model = R.RobustMVD().to(dev).train()
    s = gc.synthetic_sample(3, 64, 128, 2)
    from robustmvd_amd.registry import add_batch_dim
    im, key, po, intr, _ = add_batch_dim(s["images"], 0, s["poses"], s["intrinsics"], None)
    sample = model.input_adapter(images=im, keyview_idx=key, poses=po, intrinsics=intr)
    pred, aux = model(**sample)
    # multi-scale loss like the reference's training (the decoder detaches the up-sampled predictions between levels,
    # dispnet_decoder.py:126-138, so each head only learns from its own level's term)
    loss = sum(x.mean() for x in aux["invdepths_all"]) + sum(x.mean() for x in aux["invdepth_log_bs_all"])
    loss.backward()
    missing = [k for k, p in model.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    assert missing == [], missing
    assert float(model.encoder.conv3[0].weight.grad.abs().sum()) > 0
    assert float(model.fusion_block.corr_to_view_weight[0].weight.grad.abs().sum()) > 0
