"""Model-protocol behaviour that needs no GPU: registry, checkpoint loading in the reference's format, adapters."""
import numpy as np
import pytest
import torch

import gen_common as gc


def test_registry_and_errors():
    import robustmvd_amd as R
    assert R.list_models() == ["mvsnet_train", "robust_mvd", "robust_mvd_5M"]
    assert R.has_model("robust_mvd") and not R.has_model("nope")
    with pytest.raises(AssertionError, match="does not exist"):
        R.create_model("nope")
    with pytest.raises(RuntimeError, match="URL-only"):
        R.create_model("robust_mvd", num_gpus=0)          # pretrained=True needs the network
    with pytest.raises(ValueError, match="one process per GPU"):
        R.create_model("robust_mvd", pretrained=False, num_gpus=2)


def test_reference_format_checkpoint_loads(tmp_path):
    """{'model_state_dict': {...}} with DataParallel's 'module.' prefix, strict=True (helpers.py:146-153)"""
    import robustmvd_amd as R
    src = R.RobustMVD()
    shapes = {k: tuple(v.shape) for k, v in src.state_dict().items()}
    sd = {("module." + k): torch.from_numpy(v) for k, v in gc.robustmvd_weights(shapes, 5).items()}
    path = tmp_path / "ckpt.pt"
    torch.save({"model_state_dict": sd}, path)
    model = R.create_model("robust_mvd", weights=str(path), num_gpus=0)
    assert model.name == "robust_mvd" and callable(model.run) and not model.training
    got = model.state_dict()
    for k, v in sd.items():
        assert torch.equal(got[k[len("module."):]], v)
    assert sum(p.numel() for p in model.parameters()) == 42192877
    bad = dict(sd)
    bad.pop(next(iter(bad)))
    torch.save({"model_state_dict": bad}, path)
    with pytest.raises(RuntimeError):
        R.create_model("robust_mvd", weights=str(path), num_gpus=0)


def test_adapters_on_cpu():
    import robustmvd_amd as R
    m = R.RobustMVD().eval()
    s = gc.synthetic_sample(0, 64, 128, 2)
    from robustmvd_amd.registry import add_batch_dim
    images, key, poses, intr, _ = add_batch_dim(s["images"], 0, s["poses"], s["intrinsics"], None)
    out = m.input_adapter(images=images, keyview_idx=key, poses=poses, intrinsics=intr)
    assert out["images"][0].shape == (1, 3, 64, 128) and out["images"][0].dtype == torch.float32
    np.testing.assert_allclose(out["images"][0].numpy(), images[0] / 255.0 - 0.4, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out["intrinsics"][0][0].numpy(), intr[0][0] / np.array([[128] * 3, [64] * 3, [1] * 3], np.float32))
    # a size that is not a multiple of 64 goes through the engine's resize kernel: on a CPU model that fails loudly
    # (there is no CPU fallback); on a GPU model it works (tests/test_hip_resize.py)
    with pytest.raises(ValueError, match="needs a cuda"):
        m.input_adapter(images=[im[..., :100] for im in images], keyview_idx=key, poses=poses, intrinsics=intr)
    mv = R.MVSNet(num_sampling_steps=8).eval()
    o2 = mv.input_adapter(images=images, keyview_idx=key, poses=poses, intrinsics=intr, depth_range=(np.array([0.5]), np.array([9.0])))
    mean, std = np.array([0.485, 0.456, 0.406]).reshape(1, 3, 1, 1), np.array([0.229, 0.224, 0.225]).reshape(1, 3, 1, 1)
    np.testing.assert_allclose(o2["images"][1].numpy(), (images[1] / 255.0 - mean) / std, rtol=1e-5, atol=1e-5)
    d = mv.depth_samples(o2["depth_range"], 1, torch.device("cpu"))
    assert torch.equal(d[0], torch.linspace(0.5, 9.0, 8))
    P = mv.projection_matrices(o2["intrinsics"], o2["poses"], [0], torch.device("cpu"))
    K = intr[1][0] * np.array([[0.25] * 3, [0.25] * 3, [1.0] * 3], np.float32)
    ref = poses[1][0].copy(); ref[:3, :4] = K @ ref[:3, :4]
    np.testing.assert_allclose(P[1][0].numpy(), ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(P[0][0].numpy() @ np.block([[intr[0][0] * np.array([[0.25] * 3, [0.25] * 3, [1.0] * 3]) @ poses[0][0][:3, :4]], [poses[0][0][3:]]]),
                               np.eye(4), atol=1e-4)


def test_collate_and_index_helpers():
    from robustmvd_amd.utils import numpy_collate, select_by_index, exclude_index, to_numpy, to_torch
    a = [np.zeros((2, 3)), np.ones((2, 3)), 2 * np.ones((2, 3))]
    assert select_by_index(a, 1) is a[1]
    assert [x.sum() for x in exclude_index(a, 1)] == [0.0, 12.0]
    b = [np.stack([np.full(3, v + 10 * n) for n in range(2)]) for v in range(3)]     # views x (batch, 3)
    sel = select_by_index(b, np.array([2, 0]))
    assert sel[0][0] == 2 and sel[1][0] == 10
    exc = exclude_index(b, np.array([2, 0]))
    assert [e[0][0] for e in exc] == [0, 1] and [e[1][0] for e in exc] == [11, 12]
    col = numpy_collate([({"x": np.ones(2)}, 3, None), ({"x": np.zeros(2)}, 4, None)])
    assert col[0]["x"].shape == (2, 2) and list(col[1]) == [3, 4] and col[2] is None
    rt = to_numpy(to_torch({"a": [np.arange(3)], "b": (np.float32(1.5),)}))
    assert rt["a"][0].tolist() == [0, 1, 2]


def test_product_sampling_invdepths_match_reference_golden():
    """The PRODUCT's compute_sampling_invdepths (host code, robustmvd_amd/blocks.py) against every key of the
    reference's g1 fixture: both sampling types, S = 64 / 256, and the batched (per-sample range) form
    (planesweep_corr.py:524-555)."""
    from conftest import load_golden
    from robustmvd_amd.blocks import compute_sampling_invdepths
    g = load_golden("g1_invdepths")
    seen = set()
    for S in (64, 256):
        for typ in ("linear_invdepth", "linear_depth"):
            got = compute_sampling_invdepths(0.4, 1000.0, S, typ)
            assert tuple(got.shape) == (1, S)
            np.testing.assert_allclose(got.numpy(), g[f"S{S}_{typ}"], rtol=1e-6, atol=1e-9)
            seen.add(f"S{S}_{typ}")
    got = compute_sampling_invdepths(np.array([0.4, 0.7], np.float32), np.array([1000.0, 50.0], np.float32), 16)
    np.testing.assert_allclose(got.numpy(), g["batched_S16"], rtol=1e-6, atol=1e-9)
    seen.add("batched_S16")
    assert seen == set(g.files)
    with pytest.raises(ValueError):
        compute_sampling_invdepths(0.4, 1000.0, 8, "log")


def test_inference_only_ops_refuse_autograd():
    """the reference's sweep and fusion are differentiable.  The plain (inference) entry points of this engine
    refuse to record a graph instead of cutting the gradients silently; the operator modules route training through the
    VJP kernels (ops.*_autograd), which need the GPU like everything else: on a CPU tensor that fails loudly too."""
    import robustmvd_amd as R
    from robustmvd_amd import ops
    x = torch.zeros(1, 4, 2, 2, requires_grad=True)
    with pytest.raises(RuntimeError, match="inference-only"):
        ops.fuse_views([x, x], [x.detach(), x.detach()], [x.detach()[:, :1]] * 2)
    with pytest.raises(ValueError, match="needs a cuda"):  # differentiable path: reaches the engine, which has no CPU form
        R.PlanesweepCorrelation()(x, torch.eye(3)[None], [x], [torch.eye(4)[None]], num_sampling_points=4, min_depth=1.0,
                                  max_depth=2.0)
    with pytest.raises(RuntimeError, match="inference-only"):
        ops.softmax_regress(x, torch.zeros(1, 4))
    with torch.no_grad(), pytest.raises(ValueError, match="cuda"):  # under no_grad the same call reaches validation
        ops.softmax_regress(x, torch.zeros(1, 4))
    # robust_mvd is trainable again (K1 / K2 have backward kernels); the MVSNet path (folded BN, HIP layers) is not
    assert R.list_models(trainable_only=True) == ["robust_mvd"]


def test_inference_only_module_methods_refuse_training_mode():
    """the inference-only HIP entry points are also bound nn.Module methods (CostRegNet.forward, FeatureNet ...);
    in training mode with trainable parameters and autograd recording they must raise instead of silently returning
    graph-less outputs (the reference's modules are trainable).  Checked before any device call, so it runs on CPU."""
    import pytest
    import torch
    import robustmvd_amd as R
    net = R.CostRegNet().train()
    x = torch.zeros(1, 32, 8, 8, 8)
    with pytest.raises(RuntimeError, match="inference-only"):
        net.forward(x)
    with pytest.raises(RuntimeError, match="inference-only"):
        net.forward(x.requires_grad_())          # an input that requires grad raises as before
