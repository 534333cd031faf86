"""GPU parity tests: the HIP engine (through the C ABI) against the golden vectors of the reference's CPU
path and against the CPU oracle on identical seeded inputs.  Tolerances: SURVEY.md 8(c) — block outputs
atol = rtol = 1e-4 (fp32), masks exact up to samples within 1e-4 px of a border, regressed depth rtol 1e-3
(Path B, 11 conv layers), Path A final prediction compared in inverse-depth space at atol 1e-4."""
import numpy as np
import pytest
import torch

import gen_common as gc
from conftest import load_golden, unpack_mask
from oracle import mvd_oracle as O
from test_oracle_golden import (_sweep_inputs, check_sweep_outputs, costreg_shapes, featurenet_shapes, fusion_inputs,
                                fusion_weights)

pytestmark = pytest.mark.gpu
ATOL = RTOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    return torch.device("cuda:0")


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


@pytest.mark.parametrize("name,V", [("g2_sweep_toy", 2), ("g2_sweep_rot", 1), ("g2_sweep_behind", 1),
                                    ("g2_sweep_c256", 2), ("g2_sweep_cfg1", 1)])
def test_sweep_corr_golden(name, V, dev):
    import robustmvd_amd as R
    g = load_golden(name)
    fk, fs, Kk, Ks, Ts = _sweep_inputs(g, V)
    inv = g["invdepths"][:, :, 0, 0]
    C = fk.shape[1]
    if C % 64:  # the engine needs C % 64 == 0: zero-pad the channel axis, rescale for 1/sqrt(C)
        pad = 64 - C % 64
        fk = np.pad(fk, ((0, 0), (0, pad), (0, 0), (0, 0)))
        fs = [np.pad(f, ((0, 0), (0, pad), (0, 0), (0, 0))) for f in fs]
        rescale = np.sqrt((C + pad) / C).astype(np.float32)
    else:
        rescale = np.float32(1.0)
    blk = R.PlanesweepCorrelation()
    corrs, masks, inv_out = blk(T(fk, dev), T(Kk, dev), [T(f, dev) for f in fs], [T(t, dev) for t in Ts],
                                [T(k, dev) for k in Ks], sampling_invdepths=T(inv, dev))
    assert tuple(inv_out.shape) == g["invdepths"].shape
    check_sweep_outputs(g, V, [c.cpu().numpy() * rescale for c in corrs], [m.cpu().numpy() for m in masks])


def test_sweep_corr_range_arguments(dev):
    """(num_sampling_points, min_depth, max_depth) form == explicit invdepths; bad combinations raise."""
    import robustmvd_amd as R
    g = load_golden("g2_sweep_c256")
    fk, fs, Kk, Ks, Ts = _sweep_inputs(g, 2)
    blk = R.PlanesweepCorrelation()
    args = (T(fk, dev), T(Kk, dev), [T(f, dev) for f in fs], [T(t, dev) for t in Ts])
    c1, m1, inv = blk(*args, num_sampling_points=64, min_depth=0.4, max_depth=1000.0)
    np.testing.assert_allclose(inv.cpu().numpy(), g["invdepths"], rtol=1e-6)
    check_sweep_outputs(g, 2, [c.cpu().numpy() for c in c1], [m.cpu().numpy() for m in m1])
    with pytest.raises(ValueError):
        blk(*args, num_sampling_points=64)
    with pytest.raises(ValueError):
        blk(*args, num_sampling_points=64, min_depth=0.4, max_depth=10.0, sampling_invdepths=inv)


@pytest.mark.parametrize("V", [1, 2, 4])
def test_fusion_golden(V, dev):
    import robustmvd_amd as R
    g = load_golden("g3_fusion")
    m = R.LearnedFusion().to(dev).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in fusion_weights().items()})
    corrs, masks = fusion_inputs(V)
    with torch.no_grad():
        fused, fmask = m([T(c, dev) for c in corrs], [T(x, dev) for x in masks])
    ref = g[f"V{V}_fused"]
    assert (fmask.cpu().numpy() == unpack_mask(g[f"V{V}_fmask"], ref.shape)).all()
    # the score convs run on MIOpen (fp32): allow its accumulation-order noise on top of the block tolerance
    np.testing.assert_allclose(fused.cpu().numpy(), ref, atol=2e-4, rtol=2e-4)


@pytest.mark.parametrize("name,V", [("g4_warpvar_a", 1), ("g4_warpvar_b", 2), ("g4_warpvar_c", 2)])
@pytest.mark.parametrize("channels_last", [False, True])
def test_warp_variance_golden(name, V, channels_last, dev):
    from robustmvd_amd import ops
    g = load_golden(name)
    feats = [T(g[f"feat{i}"], dev) for i in range(V + 1)]
    projs = [T(g[f"src_proj{v}"], dev) for v in range(V)]
    var = ops.warp_variance(feats[0], feats[1:], projs, T(g["key_proj_inv"], dev), T(g["depth_values"], dev),
                            channels_last=channels_last)
    if channels_last:
        var = var.permute(0, 4, 1, 2, 3)
    np.testing.assert_allclose(var.cpu().numpy(), g["variance"], atol=ATOL, rtol=RTOL)
    if "warped0" in g.files and not channels_last:
        w0 = ops.homo_warp(feats[1], projs[0], T(g["key_proj_inv"], dev), T(g["depth_values"], dev))
        np.testing.assert_allclose(w0.cpu().numpy(), g["warped0"], atol=ATOL, rtol=RTOL)


def load_costreg(dev, seed):
    import robustmvd_amd as R
    net = R.CostRegNet().eval()
    sd = gc.fill_state_dict(costreg_shapes(), seed)
    full = net.state_dict()
    for k, v in sd.items():
        full[k] = torch.from_numpy(v)
    net.load_state_dict(full)
    return net.to(dev), sd


def test_costreg_golden(dev):
    from robustmvd_amd import ops
    from robustmvd_amd import _lib as L
    g = load_golden("g5_costreg")
    net, sd = load_costreg(dev, int(g["weight_seed"]))
    x = np.abs(gc.rng_array(int(g["x_seed"]), (1, 32, 16, 16, 24), 0.7))
    xt = T(x, dev)
    # first two layers separately (stride 1 and stride 2), then the whole U-Net
    pk = net._prepare()
    xcl = ops.to_channels_last_3d(xt)
    w, cin, cout, sc, sh, mode = pk["conv0"]
    c0 = ops.conv3d_bn_relu(xcl, w, cin, cout, sc, sh, mode)
    np.testing.assert_allclose(c0.permute(0, 4, 1, 2, 3).cpu().numpy(), g["conv0"], atol=ATOL, rtol=RTOL)
    w, cin, cout, sc, sh, mode = pk["conv1"]
    c1 = ops.conv3d_bn_relu(c0, w, cin, cout, sc, sh, mode)
    np.testing.assert_allclose(c1.permute(0, 4, 1, 2, 3).cpu().numpy(), g["conv1"], atol=ATOL, rtol=RTOL)
    out = net(xt)
    assert tuple(out.shape) == (1, 1, 16, 16, 24)
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], atol=2e-4, rtol=1e-3)


def test_featurenet_golden(dev):
    """K6 x 8 against the reference FeatureNet's outputs (g9): first layer (3-channel image input), the first stride-2
    layer, and the whole net in the reference's layout and in K3's zero-bordered staging layout."""
    import robustmvd_amd as R
    from robustmvd_amd import ops
    from robustmvd_amd import _lib as L
    g = load_golden("g9_featurenet")
    net = R.blocks.FeatureNet().eval()
    full = net.state_dict()
    for k, v in gc.fill_state_dict(featurenet_shapes(), int(g["weight_seed"])).items():
        full[k] = torch.from_numpy(v)
    net.load_state_dict(full)
    net = net.to(dev)
    x = T(gc.rng_array(int(g["x_seed"]), (2, 3, 52, 76), 0.5), dev)
    pk = net._prepare()
    w, cin, cout, k, st, sc, sh, relu = pk[0]
    c0 = ops.conv2d_bn_relu(x, w, cin, cout, k, st, sc, sh, relu=relu)
    np.testing.assert_allclose(c0.permute(0, 3, 1, 2).cpu().numpy(), g["conv0"], atol=ATOL, rtol=RTOL)
    w, cin, cout, k, st, sc, sh, relu = pk[1]
    c1 = ops.conv2d_bn_relu(c0, w, cin, cout, k, st, sc, sh, relu=relu)
    w, cin, cout, k, st, sc, sh, relu = pk[2]
    c2 = ops.conv2d_bn_relu(c1, w, cin, cout, k, st, sc, sh, relu=relu)
    np.testing.assert_allclose(c2.permute(0, 3, 1, 2).cpu().numpy(), g["conv2"], atol=ATOL, rtol=RTOL)
    out = net(x)
    assert tuple(out.shape) == (2, 32, 13, 19)
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], atol=ATOL, rtol=RTOL)
    pad = net.forward_layout(x, L.LAYOUT_NHWC_BORDER)
    assert tuple(pad.shape) == (2, 16, 22, 32)
    assert torch.equal(pad[:, 1:14, 1:20].permute(0, 3, 1, 2), out)
    border = pad.clone()
    border[:, 1:14, 1:20] = 0
    assert not border.any()


def test_channels_last_roundtrip(dev):
    from robustmvd_amd import ops
    x = torch.randn(2, 8, 5, 7, 9, device=dev)
    y = ops.to_channels_last_3d(x)
    assert torch.equal(y, x.permute(0, 2, 3, 4, 1).contiguous())
    assert torch.equal(ops.from_channels_last_3d(y), x)


@pytest.mark.parametrize("name", ["a", "b"])
def test_softmax_regress_golden(name, dev):
    from robustmvd_amd import ops
    g = load_golden("g6_regress")
    B, D, h, w = g[f"{name}_shape"]
    cost = gc.rng_array(int(g[f"{name}_seed"]), (B, D, h, w), float(g[f"{name}_scale"]))
    dv = np.stack([np.linspace(0.5, 10.0, D, dtype=np.float32)] * B)
    depth, conf = ops.softmax_regress(T(cost, dev), T(dv, dev))
    np.testing.assert_allclose(depth.cpu().numpy(), g[f"{name}_depth"], atol=1e-5, rtol=1e-5)
    # confidence depends on trunc(E[idx]); an expectation within rounding of an integer may land in the next bin
    close = np.isclose(conf.cpu().numpy(), g[f"{name}_conf"], atol=1e-5, rtol=1e-5)
    assert close.mean() > 0.999


def load_robustmvd(dev, seed):
    import robustmvd_amd as R
    model = R.RobustMVD().eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in gc.robustmvd_weights(shapes, seed).items()})
    model = model.to(dev)
    R.add_run_function(model)
    return model


def test_robustmvd_end_to_end_golden(dev):
    """create_model-protocol run() on the reference's sample_data pair at 384x576 (BASELINE config 1)."""
    g = load_golden("g7_robustmvd")
    model = load_robustmvd(dev, int(g["weight_seed"]))
    sample = dict(images=[g["image_key"].astype(np.float32), g["image_src0"].astype(np.float32)],
                  intrinsics=[g["K"].copy(), g["K"].copy()],
                  poses=[np.eye(4, dtype=np.float32), g["T0"]], keyview_idx=0)
    pred, aux = model.run(**sample)
    assert pred["depth"].shape == (1, 192, 288)
    assert model.engine_dispnet and model._engine is not None and model._engine.w is not None  # the engine's 2-D CNN ran
    # Path A's final prediction in inverse-depth space at atol 1e-4.  The 2-D convolutions run on the engine's
    # split-operand kernels here (fp32-grade: tests/test_hip_conv2d_split.py) and on oneDNN in the reference
    np.testing.assert_allclose(aux["invdepth"], g["invdepth"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(aux["invdepth_log_b"], g["invdepth_log_b"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(aux["invdepths_all"][0], g["invdepths_all_0"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(aux["invdepths_all"][3], g["invdepths_all_3"], atol=1e-4, rtol=1e-4)
    valid = g["invdepth"][0] > 1e-2
    np.testing.assert_allclose(pred["depth"][0][valid], g["depth"][0][valid], rtol=1e-3)


def test_robustmvd_two_sources_golden(dev):
    g = load_golden("g7_robustmvd_v2")
    model = load_robustmvd(dev, int(g["weight_seed"]))
    H2, W2 = 128, 192
    rng = np.random.default_rng(int(g["seed"]))
    images = [rng.uniform(0, 255, (3, H2, W2)).astype(np.float32) for _ in range(3)]
    K2 = gc.synthetic_intrinsics(H2, W2)
    poses = [gc.synthetic_pose(rng), np.eye(4, dtype=np.float32), gc.synthetic_pose(rng)]
    pred, aux = model.run(images=images, intrinsics=[K2, K2, K2], poses=poses, keyview_idx=1)
    assert model._engine is not None
    np.testing.assert_allclose(aux["invdepth"], g["invdepth"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(aux["invdepth_log_b"], g["invdepth_log_b"], atol=1e-4, rtol=1e-4)
    # the layer-by-layer form on the vendor library's convolutions (engine_dispnet=False) against the same golden output
    import robustmvd_amd as R
    lib_model = R.RobustMVD(engine_dispnet=False).eval().to(dev)
    lib_model.load_state_dict(model.state_dict())
    R.add_run_function(lib_model)
    pred_l, aux_l = lib_model.run(images=images, intrinsics=[K2, K2, K2], poses=poses, keyview_idx=1)
    assert lib_model._engine is None
    np.testing.assert_allclose(aux_l["invdepth"], g["invdepth"], atol=1e-4, rtol=1e-4)
    for k in ("invdepth", "invdepth_log_b", "invdepth_uncertainty"):
        np.testing.assert_allclose(aux[k], aux_l[k], atol=2e-5, rtol=2e-5)
    assert len(aux["invdepths_all"]) == len(aux_l["invdepths_all"]) == 6
    for a_, b_ in zip(aux["invdepths_all"], aux_l["invdepths_all"]):
        assert a_.shape == b_.shape
        np.testing.assert_allclose(a_, b_, atol=5e-5, rtol=5e-5)


def test_dispnet_conv_epilogue_is_bit_identical_to_torch(dev):
    """_ConvLeaky: conv without bias + the fused in-place bias/LeakyReLU pass equals torch's conv(bias) -> LeakyReLU."""
    from robustmvd_amd import blocks
    torch.manual_seed(3)
    for mod, shape in ((blocks.conv(3, 64, kernel_size=7, stride=2), (2, 3, 64, 96)), (blocks.conv(32, 16), (1, 32, 13, 17)),
                       (blocks._deconv(24, 12), (2, 24, 9, 11)), (blocks._iconv(14, 8), (1, 16, 10, 6))):
        mod = mod.to(dev).eval()
        x = torch.randn(*shape, device=dev)
        with torch.no_grad():
            got = mod(x)
        with torch.enable_grad():
            want = mod(x).detach()   # the plain nn.Sequential path
        assert torch.equal(got, want)


def test_frame_pipeline_matches_sequential_forwards(dev):
    """FramePipeline: frames in flight on separate streams give exactly the sequential results, in submission order."""
    import robustmvd_amd as R
    from robustmvd_amd.registry import add_batch_dim
    g = load_golden("g8_mvsnet")
    H, W, D, V = [int(v) for v in g["shape"]]
    model = R.MVSNet(num_sampling_steps=D).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    full = model.state_dict()
    for k, v in gc.fill_state_dict(shapes, int(g["weight_seed"])).items():
        full[k] = torch.from_numpy(v)
    model.load_state_dict(full)
    model = model.to(dev)

    def sample(seed):
        s = gc.synthetic_sample(seed, H, W, V)
        im, key, po, intr, dr = add_batch_dim(s["images"], 0, s["poses"], s["intrinsics"], (np.float32(0.5), np.float32(10.0)))
        return model.input_adapter(images=im, keyview_idx=key, poses=po, intrinsics=intr, depth_range=dr)

    samples = [sample(int(g["sample_seed"]) + i) for i in range(5)]
    with torch.no_grad():
        want = [model(**s)[0]["depth"].clone() for s in samples]
    pipe = R.FramePipeline(model, depth=3)
    tickets = [pipe.submit(**s) for s in samples]
    got = [t.result()[0]["depth"] for t in tickets]
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    np.testing.assert_allclose(got[0].cpu().numpy(), g["depth"], rtol=1e-3)
    with pytest.raises(ValueError):
        R.FramePipeline(model, depth=0)


def test_frame_pipeline_cold_start(dev):
    """a FRESH model (no forward yet, so no packed weights) handed straight to FramePipeline: the lazy weight
    packing must not race between the side streams.  Results equal sequential forwards run afterwards."""
    import robustmvd_amd as R
    from robustmvd_amd.registry import add_batch_dim
    H, W, D, V = 64, 96, 16, 2
    model = R.MVSNet(num_sampling_steps=D).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    full = model.state_dict()
    for k, v in gc.fill_state_dict(shapes, 3).items():
        full[k] = torch.from_numpy(v)
    model.load_state_dict(full)
    model = model.to(dev)
    samples = []
    for i in range(4):
        s = gc.synthetic_sample(70 + i, H, W, V)
        im, key, po, intr, dr = add_batch_dim(s["images"], 0, s["poses"], s["intrinsics"], (np.float32(0.5), np.float32(10.0)))
        samples.append(model.input_adapter(images=im, keyview_idx=key, poses=po, intrinsics=intr, depth_range=dr))
    assert model.feature._packed is None and model.cost_regularization._packed is None
    pipe = R.FramePipeline(model, depth=3)
    assert model.feature._packed is not None and model.cost_regularization._packed is not None
    got = [t.result()[0]["depth"] for t in [pipe.submit(**s) for s in samples]]
    with torch.no_grad():
        want = [model(**s)[0]["depth"] for s in samples]
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_input_adapters_normalise_on_device_like_numpy(dev):
    """the adapters upload raw images and normalise on the GPU: bit-identical to the reference's numpy arithmetic
    (rmvd/models/mvsnet.py:181-183, robust_mvd.py:113-116)"""
    import robustmvd_amd as R
    rng = np.random.default_rng(5)
    images = [rng.uniform(0, 255, (1, 3, 64, 128)).astype(np.float32) for _ in range(2)]
    K = gc.synthetic_intrinsics(64, 128)[None]
    poses = [np.eye(4, dtype=np.float32)[None]] * 2
    mvs = R.MVSNet(num_sampling_steps=8).to(dev)
    got = mvs.input_adapter(images=images, keyview_idx=np.array([0]), poses=poses, intrinsics=[K, K], depth_range=(np.array([0.5], np.float32), np.array([10.0], np.float32)))
    # the reference's chain, operation for operation (transforms.py:283-311): float32 /255, then a subtraction and a
    # division by Python LISTS on an (N,H,W,3) view -- float64 arithmetic -- then astype(float32)
    shift, scale = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    for im, g_ in zip(images, got["images"]):
        x = np.transpose(im / 255.0, [0, 2, 3, 1])
        assert x.dtype == np.float32
        want = np.transpose(((x - shift) / scale), [0, 3, 1, 2]).astype(np.float32)
        assert g_.is_cuda and g_.dtype == torch.float32 and np.array_equal(g_.cpu().numpy(), want)
    rm = R.RobustMVD().to(dev)
    got = rm.input_adapter(images=images, keyview_idx=np.array([0]), poses=poses, intrinsics=[K, K])
    for im, g_ in zip(images, got["images"]):
        assert np.array_equal(g_.cpu().numpy(), (im / 255.0 - 0.4).astype(np.float32))


def test_mvsnet_end_to_end_golden(dev):
    import robustmvd_amd as R
    g = load_golden("g8_mvsnet")
    H, W, D, V = g["shape"]
    model = R.MVSNet(num_sampling_steps=int(D)).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    full = model.state_dict()
    for k, v in gc.fill_state_dict(shapes, int(g["weight_seed"])).items():
        full[k] = torch.from_numpy(v)
    model.load_state_dict(full)
    model = model.to(dev)
    R.add_run_function(model)
    s = gc.synthetic_sample(int(g["sample_seed"]), int(H), int(W), int(V))
    pred, _ = model.run(images=s["images"], poses=s["poses"], intrinsics=s["intrinsics"], keyview_idx=0,
                        depth_range=(np.float32(0.5), np.float32(10.0)))
    np.testing.assert_allclose(pred["depth"][None], g["depth"], rtol=1e-3)   # Path B depth rtol 1e-3
    np.testing.assert_allclose(pred["depth_uncertainty"][None], g["depth_uncertainty"], atol=2e-3)


@pytest.mark.parametrize("name,norm", [("none", False), ("before", "before"), ("pp", "dim")])
def test_sweep_block_options_golden(name, norm, dev):
    """PlanesweepCorrelation(normalize=False / "before") and per-key-pixel sampling inverse depths (N,S,H,W) against
    the reference block's outputs (g12)."""
    import robustmvd_amd as R
    g = load_golden("g12_sweep_options")
    fk = T(gc.rng_array(1401, (1, 64, 12, 18)), dev)
    fs = [T(gc.rng_array(1402 + i, (1, 64, 12, 18)), dev) for i in range(2)]
    blk = R.PlanesweepCorrelation(normalize=norm)
    kw = dict(sampling_invdepths=T(g["invdepths_pp"], dev)) if name == "pp" else dict(num_sampling_points=8, min_depth=0.4, max_depth=1000.0)
    corrs, masks, inv = blk(fk, T(g["K"], dev), fs, [T(g["T0"], dev), T(g["T1"], dev)], **kw)
    assert tuple(inv.shape) == ((1, 8, 12, 18) if name == "pp" else (1, 8, 1, 1))
    for v in range(2):
        ref_c = g[f"{name}_corr{v}"]
        ref_m = unpack_mask(g[f"{name}_mask{v}"], ref_c.shape)
        m = masks[v].cpu().numpy()
        assert (m != ref_m).mean() <= 1e-3
        ok = m == ref_m
        np.testing.assert_allclose(corrs[v].cpu().numpy()[ok], ref_c[ok], atol=ATOL, rtol=RTOL)


@pytest.mark.gpu
@pytest.mark.parametrize("name,norm", [("none", False), ("before", "before"), ("after", True)])
def test_warp_only_block_golden(name, norm, dev):
    """PlanesweepCorrelation(warp_only=True) (WarpOnlyCorr, planesweep_corr.py:107-140) against the reference block's
    outputs (g13), plus what the reference cannot run: sources of another size than the key map (against the oracle)."""
    import robustmvd_amd as R
    g = load_golden("g13_warp_only")
    fk = T(gc.rng_array(1501, (1, 16, 12, 18)), dev)
    fs_np = [gc.rng_array(1502, (1, 16, 12, 18)), gc.rng_array(1503, (1, 16, 12, 18))]
    blk = R.PlanesweepCorrelation(warp_only=True, normalize=norm)
    Ts = [T(g["T0"], dev), T(g["T1"], dev)]
    warped, masks, inv = blk(fk, T(g["K"], dev), [T(f, dev) for f in fs_np], Ts, num_sampling_points=6, min_depth=0.4, max_depth=1000.0)
    np.testing.assert_allclose(inv.cpu().numpy(), g["invdepths"], rtol=1e-6)
    for v in range(2):
        ref_w = g[f"{name}_warped{v}"]
        ref_m = unpack_mask(g[f"{name}_mask{v}"], (1, 6, 12, 18))
        m = masks[v].cpu().numpy()
        assert tuple(warped[v].shape) == (1, 6, 16, 12, 18)
        assert (m != ref_m).mean() <= 1e-3
        ok = np.broadcast_to((m == ref_m)[:, :, None], ref_w.shape)
        np.testing.assert_allclose(warped[v].cpu().numpy()[ok], ref_w[ok], atol=ATOL, rtol=RTOL)
    # a smaller source map, two batch elements with their own inverse depths
    rng = np.random.default_rng(5)
    fs2 = [rng.standard_normal((2, 16, 10, 14)).astype(np.float32)]
    K2 = np.repeat(g["K"], 2, 0); T2 = np.repeat(g["T1"], 2, 0)
    inv2 = np.stack([g["invdepths"].reshape(-1), g["invdepths"].reshape(-1) * 0.7]).astype(np.float32)
    w_o, m_o = O.planesweep_warp((12, 18), K2, fs2, [T2], inv2, normalize=norm)
    w_h, m_h, _ = blk(T(np.zeros((2, 16, 12, 18), np.float32), dev), T(K2, dev), [T(fs2[0], dev)], [T(T2, dev)], sampling_invdepths=T(inv2, dev))
    mh = m_h[0].cpu().numpy()
    assert (mh != m_o[0]).mean() <= 1e-3
    ok = np.broadcast_to((mh == m_o[0])[:, :, None], w_o[0].shape)
    np.testing.assert_allclose(w_h[0].cpu().numpy()[ok], w_o[0][ok], atol=ATOL, rtol=RTOL)
    with pytest.raises(ValueError):
        blk(fk, T(g["K"], dev), [T(fs_np[0], dev).requires_grad_()], Ts[:1], num_sampling_points=6, min_depth=0.4, max_depth=1000.0)


def test_pinned_uploader_matches_direct_upload(dev):
    """PinnedUploader (pinned host buffers + copy stream): what arrives on the device is what a plain .to() uploads, and a
    forward fed from it equals the forward on directly uploaded images."""
    import robustmvd_amd as R
    rng = np.random.default_rng(3)
    images = [rng.uniform(0, 255, (1, 3, 64, 96)).astype(np.float32) for _ in range(3)]
    up = R.PinnedUploader(dev)
    staged = [up.stage(images) for _ in range(3)]  # several frames in flight on the copy stream
    for st in staged:
        got = st.wait()
        for a, b in zip(got, images):
            assert a.is_cuda and torch.equal(a.cpu(), torch.from_numpy(b))
    with pytest.raises(ValueError):
        R.PinnedUploader("cpu")


def test_pinned_uploader_consumed_on_another_stream(dev):
    """tensors staged by PinnedUploader live in the copy stream's allocator pool; wait() must record the CONSUMING
    stream (not the stream current at stage() time), or the next stage() could recycle the block under a reader.  Stage on the
    default stream, consume on a side stream while further frames are staged, compare with the host data."""
    import robustmvd_amd as R
    rng = np.random.default_rng(3)
    frames = [[rng.uniform(0, 255, (3, 96, 128)).astype(np.float32) for _ in range(3)] for _ in range(6)]
    up = R.PinnedUploader(dev)
    side = torch.cuda.Stream(dev)
    sums, nxt = [], up.stage(frames[0])
    for i in range(len(frames)):
        cur, nxt = nxt, up.stage(frames[(i + 1) % len(frames)])
        with torch.cuda.stream(side):
            imgs = cur.wait()
            acc = torch.zeros((), device=dev, dtype=torch.float64)
            for _ in range(20):                      # keep the side stream busy while the next frames are staged
                acc = acc + sum(im.double().sum() for im in imgs) / 20
            sums.append(acc)
        del cur, imgs
    torch.cuda.synchronize(dev)
    for i, s_ in enumerate(sums):
        want = sum(float(np.asarray(im, np.float64).sum()) for im in frames[i])
        assert abs(float(s_) - want) <= 1e-9 * abs(want)


def test_robustmvd_engine_matches_vendor_library_form_batch_two_three_sources(dev):
    """The engine's 2-D CNN (dispnet_engine.py) against the layer-by-layer form on the vendor library's convolutions: batch of 2,
    3 source views, key view in the middle; and the cases that must fall back to the layer-by-layer form."""
    import robustmvd_amd as R
    torch.manual_seed(11)
    eng = R.RobustMVD().eval().to(dev)
    lib = R.RobustMVD(engine_dispnet=False).eval().to(dev)
    lib.load_state_dict(eng.state_dict())
    n, H, W = 2, 128, 192
    rng = np.random.default_rng(5)
    images = [torch.from_numpy(rng.uniform(-0.4, 0.6, (n, 3, H, W)).astype(np.float32)).to(dev) for _ in range(4)]
    K = gc.synthetic_intrinsics(H, W) / np.array([[W] * 3, [H] * 3, [1.0] * 3], np.float32)
    intr = [torch.from_numpy(np.stack([K] * n).astype(np.float32)).to(dev) for _ in range(4)]
    poses = [torch.from_numpy(np.stack([gc.synthetic_pose(rng) for _ in range(n)])).to(dev) for _ in range(4)]
    poses[2] = torch.eye(4, device=dev).repeat(n, 1, 1)
    key = torch.tensor([2, 2])
    with torch.no_grad():
        pe, ae = eng(images=images, poses=poses, intrinsics=intr, keyview_idx=key)
        pl, al = lib(images=images, poses=poses, intrinsics=intr, keyview_idx=key)
    assert eng._engine is not None and lib._engine is None
    assert pe["depth"].shape == pl["depth"].shape == (n, 1, H // 2, W // 2)
    for k in ("invdepth", "invdepth_log_b", "invdepth_uncertainty"):
        torch.testing.assert_close(ae[k], al[k], atol=3e-5, rtol=3e-5)
    for a_, b_ in zip(ae["invdepths_all"], al["invdepths_all"]):
        torch.testing.assert_close(a_, b_, atol=1e-4, rtol=1e-4)
    # one source view (LearnedFusion passes it through)
    with torch.no_grad():
        p1, a1 = eng(images=images[1:3], poses=poses[1:3], intrinsics=intr[1:3], keyview_idx=torch.tensor([1, 1]))
        p2, a2 = lib(images=images[1:3], poses=poses[1:3], intrinsics=intr[1:3], keyview_idx=torch.tensor([1, 1]))
    torch.testing.assert_close(a1["invdepth"], a2["invdepth"], atol=3e-5, rtol=3e-5)
    # under autograd the layer-by-layer form runs (the engine is inference only)
    eng._engine = None
    p3, a3 = eng(images=images, poses=poses, intrinsics=intr, keyview_idx=key)
    p4, a4 = lib(images=images, poses=poses, intrinsics=intr, keyview_idx=key)
    assert eng._engine is None and a3["invdepth"].requires_grad
    torch.testing.assert_close(a3["invdepth"], a4["invdepth"], atol=1e-5, rtol=1e-5)
