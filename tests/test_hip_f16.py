"""GPU parity of the fp16-feature variant (BASELINE.json configs[3]: 896x1216, 4 source views, 256 planes, "3D-conv
regulariser on MFMA, fp16 features"; SURVEY.md 8b `mvd_warp_variance_{f32,f16}`).

What is fp16: the feature maps handed to the sweep (one rounding), the stored variance volume (one rounding), and the
operands of the regulariser's first layer (fp16 MFMA, fp32 accumulation).  Everything else is the fp32 path.
  * K3-f16 against the fp32 kernel / the CPU oracle fed the SAME fp16-representable features: identical arithmetic, so the
    only difference is the final rounding of the volume to fp16 (rel 2^-11);
  * conv0-f16 against the C oracle's conv3d on the fp16-rounded input and weights: fp32 accumulation order differs
    (one K=32 MFMA per tap) -> atol/rtol 1e-3;
  * the whole MVSNet forward with half_features=True against the fp32 CPU oracle pipeline at configs[3]'s full size:
    regressed depth rtol 1e-2.
"""
import numpy as np
import pytest
import torch

import gen_common as gc
from oracle import c_oracle as CO
from oracle import pipeline as PL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def bordered(f, dev, dtype):
    """(B,C,h,w) numpy -> zero-bordered channel-last (B,h+3,w+3,C) device tensor"""
    B, C, h, w = f.shape
    out = torch.zeros((B, h + 3, w + 3, C), dtype=dtype, device=dev)
    out[:, 1:h + 1, 1:w + 1] = T(f, dev).permute(0, 2, 3, 1).to(dtype)
    return out


@pytest.mark.parametrize("B,h,w,D,V", [(1, 24, 40, 8, 2), (2, 37, 53, 5, 3), (1, 9, 70, 7, 6), (1, 64, 96, 16, 4),
                                       (1, 21, 30, 11, 20), (1, 13, 22, 17, 12), (1, 10, 19, 9, 16)])
def test_warp_variance_f16_equals_f32_kernel_on_fp16_features(B, h, w, D, V, dev):
    from robustmvd_amd import ops
    from test_hip_shapes import mvs_inputs
    feats, projs, key_inv, depth = mvs_inputs(B, 32, h, w, D, V, seed=h * 7 + V)
    feats = [f.astype(np.float16).astype(np.float32) for f in feats]  # fp16-representable values
    args = ([T(p, dev) for p in projs], T(key_inv, dev), T(depth, dev))
    f32 = [bordered(f, dev, torch.float32) for f in feats]
    f16 = [bordered(f, dev, torch.float16) for f in feats]
    want = ops.warp_variance(f32[0], f32[1:], *args, channels_last=True, staged=True)
    got = ops.warp_variance_f16(f16[0], f16[1:], *args)
    assert got.dtype == torch.float16 and tuple(got.shape) == (B, D, h, w, 32)
    assert torch.equal(got, want.half())  # same fp32 arithmetic, one rounding at the end
    ref = CO.warp_variance(feats[0], feats[1:], projs, key_inv, depth)
    np.testing.assert_allclose(got.float().permute(0, 4, 1, 2, 3).cpu().numpy(), ref, atol=2e-3, rtol=2e-3)
    with pytest.raises(ValueError, match="float16"):
        ops.warp_variance_f16(f32[0], f32[1:], *args)


def test_convert_roundtrip(dev):
    from robustmvd_amd import ops
    x = torch.randn(3, 5, 8, device=dev) * 100
    assert torch.equal(ops.to_f16(x), x.half())


@pytest.mark.parametrize("D,h,w", [(5, 7, 50), (9, 12, 130), (33, 6, 64), (4, 17, 16)])
def test_conv0_f16_vs_oracle(D, h, w, dev):
    from robustmvd_amd import ops
    rng = np.random.default_rng(D * 100 + w)
    x = rng.standard_normal((32, D, h, w)).astype(np.float16)
    wt = (rng.standard_normal((8, 32, 3, 3, 3)) * np.sqrt(2 / (32 * 27))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, 8).astype(np.float32)
    shift = (rng.standard_normal(8) * 0.1).astype(np.float32)
    ref = CO.conv3d(x.astype(np.float32), wt.astype(np.float16).astype(np.float32), scale, shift, stride=1, relu=True)
    xt = T(x, dev).permute(1, 2, 3, 0).contiguous()[None]
    got = ops.conv3d_bn_relu_f16in(xt, ops.pack_conv3d_weights_f16(T(wt, dev)), T(scale, dev), T(shift, dev))
    assert got.dtype == torch.float32 and tuple(got.shape) == (1, D, h, w, 8)
    np.testing.assert_allclose(got[0].permute(3, 0, 1, 2).cpu().numpy(), ref, atol=1e-3, rtol=1e-3)


def _half_model(D, seed, dev):
    import robustmvd_amd as R
    model = R.MVSNet(num_sampling_steps=D, half_features=True).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = gc.fill_state_dict(shapes, seed)
    full = model.state_dict()
    for k, v in sd.items():
        full[k] = torch.from_numpy(v)
    model.load_state_dict(full)
    return R.add_run_function(model.to(dev)), sd


def _check_against_fp32_oracle(H, W, V, D, seed, dev):
    model, sd = _half_model(D, seed, dev)
    s = gc.synthetic_sample(seed, H, W, V)
    pred, _ = model.run(images=s["images"], poses=s["poses"], intrinsics=s["intrinsics"], keyview_idx=0,
                        depth_range=(np.float32(0.5), np.float32(10.0)))
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(3, 1, 1)
    images = [((im / 255.0 - mean) / std).astype(np.float32)[None] for im in s["images"]]
    ref = PL.mvsnet_forward(images, [p[None] for p in s["poses"]], [k[None] for k in s["intrinsics"]], 0, (0.5, 10.0), sd, D)
    np.testing.assert_allclose(pred["depth"], ref["depth"][0], rtol=1e-2)  # fp16-feature config
    return pred, ref


def test_mvsnet_half_features_small_vs_fp32_oracle(dev):
    _check_against_fp32_oracle(64, 96, 2, 16, 21, dev)


def test_mvsnet_half_features_at_config3_vs_fp32_oracle(dev):
    """BASELINE configs[3] as named: 896x1216, 4 source views, 256 planes -> 224x304x256 volume, fp16 features."""
    pred, ref = _check_against_fp32_oracle(896, 1216, 4, 256, 103, dev)
    assert pred["depth"].shape == (1, 224, 304)
    rel = np.abs(pred["depth"] - ref["depth"][0]) / ref["depth"][0]
    print(f"configs[3] half_features: max rel depth error {rel.max():.2e}, median {np.median(rel):.2e}")


@pytest.mark.parametrize("D,h,w", [(5, 7, 50), (9, 12, 130), (33, 6, 64), (4, 17, 16)])
def test_conv0_split_operands_vs_fp32_kernel_and_oracle(D, h, w, dev):
    """split-operand conv0 (two fp16 terms per fp32 operand, fp16 MFMA, fp32 accumulate) against the fp32-MFMA kernel on
    the same fp32 data: fp32-grade agreement (the dropped a_lo*w_lo term is 2^-22 relative), and against
    the C oracle at the block tolerance."""
    from robustmvd_amd import ops, _lib as L
    rng = np.random.default_rng(D * 7 + w)
    x = (rng.standard_normal((32, D, h, w)) * rng.choice([1e-3, 1.0, 30.0], size=(32, 1, 1, 1))).astype(np.float32)
    wt = (rng.standard_normal((8, 32, 3, 3, 3)) * np.sqrt(2 / (32 * 27))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, 8).astype(np.float32)
    shift = (rng.standard_normal(8) * 0.1).astype(np.float32)
    xt = T(x, dev).permute(1, 2, 3, 0).contiguous()[None]
    got = ops.conv3d_bn_relu_split(xt, ops.pack_conv3d_weights_split(T(wt, dev)), T(scale, dev), T(shift, dev))
    w32, _, _ = ops.pack_conv3d_weights(T(wt, dev), L.CONV3D_STRIDE1)
    base = ops.conv3d_bn_relu(xt, w32, 32, 8, T(scale, dev), T(shift, dev), L.CONV3D_STRIDE1, relu=True)
    ref = CO.conv3d(x, wt, scale, shift, stride=1, relu=True)
    g, b_ = got[0].permute(3, 0, 1, 2).cpu().numpy(), base[0].permute(3, 0, 1, 2).cpu().numpy()
    mag = np.abs(ref).max()
    assert np.abs(g - b_).max() <= 4e-6 * mag, np.abs(g - b_).max() / mag      # fp32-grade: both are ~1e-6 from the exact sum
    assert np.abs(g.astype(np.float64) - ref).max() <= 2.0 * max(np.abs(b_.astype(np.float64) - ref).max(), 1e-6 * mag)
    np.testing.assert_allclose(g, ref, atol=1e-4 * max(mag, 1.0), rtol=1e-4)


def conv3d_f64(x, wt):
    """float64 reference: x (32,D,h,w), wt (8,32,3,3,3), stride 1, padding 1 -> (8,D,h,w); non-finite inputs propagate"""
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (1, 1)))
    D, h, w = x.shape[1:]
    y = np.zeros((8, D, h, w), np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        for kd in range(3):
            for kh in range(3):
                for kw in range(3):
                    y += np.einsum("oc,cdhw->odhw", wt[:, :, kd, kh, kw].astype(np.float64), xp[:, kd:kd + D, kh:kh + h, kw:kw + w])
    return y


@pytest.mark.parametrize("mag", [1e-42, 1e-30, 1e-12, 1e-3, 1.0, 3e4, 1e5, 1e12, 1e30])
@pytest.mark.parametrize("wmag", [1.0, 1e-7, 1e6])
def test_conv0_split_error_vs_float64_across_magnitudes(mag, wmag, dev):
    """the emulated first layer may be the default only if it is fp32-grade over the whole fp32 range.
    Against a float64 convolution of the same fp32 data: max |error| of the split kernel <= 1.5 x max |error| of the
    fp32-MFMA kernel, for activations of magnitude 1e-42 (denormals) .. 1e30 and weights of magnitude 1e-7 .. 1e6 (the
    power-of-two range scaling makes the error independent of both; without it fp16 overflows beyond 65504)."""
    from robustmvd_amd import ops, _lib as L
    if mag * wmag > 1e33:
        pytest.skip("the exact result exceeds fp32's range (inf on either kernel)")
    rng = np.random.default_rng(int(abs(np.log10(mag)) * 10 + abs(np.log10(wmag))))
    D, h, w = 6, 9, 40
    x = (np.abs(rng.standard_normal((32, D, h, w))) * rng.choice([1e-3, 1.0, 30.0], size=(32, 1, 1, 1)) * mag).astype(np.float32)
    wt = (rng.standard_normal((8, 32, 3, 3, 3)) * np.sqrt(2 / (32 * 27)) * wmag).astype(np.float32)
    wt[3] *= 64.0  # per-channel weight ranges differ
    one, zero = np.ones(8, np.float32), np.zeros(8, np.float32)
    xt = T(x, dev).permute(1, 2, 3, 0).contiguous()[None]
    got = ops.conv3d_bn_relu_split(xt, ops.pack_conv3d_weights_split(T(wt, dev)), T(one, dev), T(zero, dev), relu=False)
    w32, _, _ = ops.pack_conv3d_weights(T(wt, dev), L.CONV3D_STRIDE1)
    base = ops.conv3d_bn_relu(xt, w32, 32, 8, T(one, dev), T(zero, dev), L.CONV3D_STRIDE1, relu=False)
    ref = conv3d_f64(x, wt)
    g = got[0].permute(3, 0, 1, 2).cpu().numpy().astype(np.float64)
    b_ = base[0].permute(3, 0, 1, 2).cpu().numpy().astype(np.float64)
    assert np.isfinite(g).all()
    err_s, err_f = np.abs(g - ref).max(), np.abs(b_ - ref).max()
    print(f"mag {mag:g} wmag {wmag:g}: split {err_s / np.abs(ref).max():.2e}  fp32-MFMA {err_f / np.abs(ref).max():.2e} (relative to max |y|)")
    assert err_s <= 1.5 * err_f, (err_s, err_f)


@pytest.mark.parametrize("cin,cout,D,h,w", [(32, 32, 8, 12, 40), (32, 16, 5, 7, 50), (32, 64, 4, 9, 33), (16, 16, 8, 12, 40), (16, 8, 5, 7, 50),
                                            (16, 32, 9, 6, 64), (16, 16, 3, 5, 31)])
def test_conv_split_more_output_channels_vs_oracle_and_float64(cin, cout, D, h, w, dev):
    """the split-operand kernel with 16 / 32 / 64 output channels (8 per workgroup; conv4 of the regulariser is 32 -> 32):
    against the C oracle at the block tolerance, and against float64 no worse than 1.5 x the fp32-MFMA kernel"""
    from robustmvd_amd import ops, _lib as L
    rng = np.random.default_rng(cout + w + cin)
    x = np.abs(rng.standard_normal((cin, D, h, w)) * rng.choice([1e-2, 1.0, 20.0], size=(cin, 1, 1, 1))).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3, 3)) * np.sqrt(2 / (cin * 27))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    xt = T(x, dev).permute(1, 2, 3, 0).contiguous()[None]
    got = ops.conv3d_bn_relu_split(xt, ops.pack_conv3d_weights_split(T(wt, dev)), T(scale, dev), T(shift, dev))
    assert tuple(got.shape) == (1, D, h, w, cout)
    g = got[0].permute(3, 0, 1, 2).cpu().numpy()
    ref = CO.conv3d(x, wt, scale, shift, stride=1, relu=True)
    np.testing.assert_allclose(g, ref, atol=1e-4 * max(np.abs(ref).max(), 1.0), rtol=1e-4)
    w32, _, _ = ops.pack_conv3d_weights(T(wt, dev), L.CONV3D_STRIDE1)
    one, zero = np.ones(cout, np.float32), np.zeros(cout, np.float32)
    a = ops.conv3d_bn_relu_split(xt, ops.pack_conv3d_weights_split(T(wt, dev)), T(one, dev), T(zero, dev), relu=False)
    b_ = ops.conv3d_bn_relu(xt, w32, cin, cout, T(one, dev), T(zero, dev), L.CONV3D_STRIDE1, relu=False)
    r64 = np.zeros((cout, D, h, w))
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (1, 1)))
    for kd in range(3):
        for kh in range(3):
            for kw in range(3):
                r64 += np.einsum("oc,cdhw->odhw", wt[:, :, kd, kh, kw].astype(np.float64), xp[:, kd:kd + D, kh:kh + h, kw:kw + w])
    ea = np.abs(a[0].permute(3, 0, 1, 2).cpu().numpy() - r64).max()
    eb = np.abs(b_[0].permute(3, 0, 1, 2).cpu().numpy() - r64).max()
    assert ea <= 1.5 * eb, (ea, eb)


def test_conv0_split_nonfinite_and_mixed_magnitudes(dev):
    """inf / NaN activations reach the same outputs as on the fp32 kernel (non-finite there, untouched elsewhere), and a volume
    whose magnitudes span 12 decades keeps fp32-grade accuracy where it matters: the error of every output stays within 1.5 x
    the fp32 kernel's worst error"""
    from robustmvd_amd import ops, _lib as L
    rng = np.random.default_rng(5)
    D, h, w = 6, 9, 40
    x = (np.abs(rng.standard_normal((32, D, h, w))) * 10.0 ** rng.uniform(-6, 6, size=(1, D, h, w))).astype(np.float32)
    wt = (rng.standard_normal((8, 32, 3, 3, 3)) * np.sqrt(2 / (32 * 27))).astype(np.float32)
    one, zero = np.ones(8, np.float32), np.zeros(8, np.float32)
    pk, (w32, _, _) = ops.pack_conv3d_weights_split(T(wt, dev)), ops.pack_conv3d_weights(T(wt, dev), L.CONV3D_STRIDE1)

    def both(xa):
        xt = T(xa, dev).permute(1, 2, 3, 0).contiguous()[None]
        a = ops.conv3d_bn_relu_split(xt, pk, T(one, dev), T(zero, dev), relu=False)[0].permute(3, 0, 1, 2).cpu().numpy()
        b_ = ops.conv3d_bn_relu(xt, w32, 32, 8, T(one, dev), T(zero, dev), L.CONV3D_STRIDE1, relu=False)[0].permute(3, 0, 1, 2).cpu().numpy()
        return a.astype(np.float64), b_.astype(np.float64)

    g, b_ = both(x)
    ref = conv3d_f64(x, wt)
    assert np.abs(g - ref).max() <= 1.5 * np.abs(b_ - ref).max()
    xi = x.copy()
    xi[5, 2, 4, 7] = np.inf
    xi[9, 4, 1, 30] = np.nan
    g, b_ = both(xi)
    # non-finite exactly where the exact convolution is: the 3x3x3 neighbourhoods of the two voxels, all 8 channels.  (The
    # fp32 kernel's set is a superset: its paired MFMA rows multiply a neighbour's inf by a zero weight.)
    bad = ~np.isfinite(conv3d_f64(xi, wt))
    assert bad.any() and (~np.isfinite(g) == bad).all() and (~np.isfinite(b_))[bad].all()
    refi = conv3d_f64(np.where(np.isfinite(xi), xi, 0), wt)
    ok = np.isfinite(b_)
    # inf and NaN do not take part in max |x| (the activation scale): every other output keeps its accuracy
    assert np.abs(g[ok] - refi[ok]).max() <= 1.5 * max(np.abs(b_[ok] - refi[ok]).max(), 1e-7 * np.abs(refi[ok]).max())


def test_mvsnet_fp32_conv0_matches_default_model(dev):
    """MVSNet(conv0_split=False) (first regulariser layer on fp32 MFMA) against the default model (split operands, range
    scaled) on the same weights and inputs: regressed depth within 1e-5."""
    import robustmvd_amd as R
    H, W, V, D = 128, 192, 2, 32
    m0 = R.MVSNet(num_sampling_steps=D).eval()
    shapes = {k: tuple(v.shape) for k, v in m0.state_dict().items()}
    sd = gc.fill_state_dict(shapes, 8)
    full = m0.state_dict()
    for k, v in sd.items():
        full[k] = torch.from_numpy(v)
    m0.load_state_dict(full)
    m1 = R.MVSNet(num_sampling_steps=D, conv0_split=False).eval()
    m1.load_state_dict(full)
    m0, m1 = R.add_run_function(m0.to(dev)), R.add_run_function(m1.to(dev))
    s = gc.synthetic_sample(4, H, W, V)
    kw = dict(images=s["images"], poses=s["poses"], intrinsics=s["intrinsics"], keyview_idx=0, depth_range=(np.float32(0.5), np.float32(10.0)))
    p0, _ = m0.run(**kw)
    p1, _ = m1.run(**kw)
    # two fp32-grade evaluations of the same network (FeatureNet's and the regulariser's stride-1 / stride-2 layers on the split-operand
    # kernels against every layer on the fp32 matrix instruction): they differ by a few fp32 roundings of the regressed depth
    np.testing.assert_allclose(p1["depth"], p0["depth"], rtol=3e-5)


def test_conv0_f16_and_split_batch_of_two(dev):
    """both MFMA-fp16 forms of conv0 on TWO batch elements with a ragged tile grid and two plane groups per element
    (the batch index is the slowest part of their block decode), against the C oracle per element"""
    from robustmvd_amd import ops
    rng = np.random.default_rng(77)
    B, D, h, w = 2, 40, 10, 70
    wt = (rng.standard_normal((8, 32, 3, 3, 3)) * np.sqrt(2 / (32 * 27))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, 8).astype(np.float32)
    shift = (rng.standard_normal(8) * 0.1).astype(np.float32)
    x = rng.standard_normal((B, 32, D, h, w)).astype(np.float32)
    xt = T(x, dev).permute(0, 2, 3, 4, 1).contiguous()
    got = ops.conv3d_bn_relu_split(xt, ops.pack_conv3d_weights_split(T(wt, dev)), T(scale, dev), T(shift, dev))
    ref = np.stack([CO.conv3d(x[b], wt, scale, shift, stride=1, relu=True) for b in range(B)])
    np.testing.assert_allclose(got.permute(0, 4, 1, 2, 3).cpu().numpy(), ref, atol=1e-4, rtol=1e-4)
    x16 = x.astype(np.float16)
    got16 = ops.conv3d_bn_relu_f16in(T(x16, dev).permute(0, 2, 3, 4, 1).contiguous(), ops.pack_conv3d_weights_f16(T(wt, dev)),
                                     T(scale, dev), T(shift, dev))
    ref16 = np.stack([CO.conv3d(x16[b].astype(np.float32), wt.astype(np.float16).astype(np.float32), scale, shift, stride=1, relu=True)
                      for b in range(B)])
    np.testing.assert_allclose(got16.permute(0, 4, 1, 2, 3).cpu().numpy(), ref16, atol=1e-3, rtol=1e-3)


def test_fp16_variant_range_is_plain_ieee(dev):
    """the fp16-feature variant does no range handling; pin what happens outside fp16's range: saturation to inf
    above 65504, flush below the smallest subnormal, NaN kept (include/mvd.h states it)."""
    from robustmvd_amd import ops
    x = torch.tensor([1.0, 65504.0, 65520.0, 1e5, -1e6, 6e-8, 2e-8, float("nan")], device=dev)
    y = ops.to_f16(x).float().cpu().numpy()
    assert y[0] == 1.0 and y[1] == 65504.0 and np.isinf(y[2]) and np.isinf(y[3]) and y[4] == -np.inf
    assert y[5] == np.float32(np.float16(6e-8)) and y[6] == 0.0 and np.isnan(y[7])


def _finite_absmax(t):
    a = t.abs().flatten()
    a = a[torch.isfinite(a)]
    return float(a.max()) if a.numel() else 0.0


@pytest.mark.parametrize("mode_name,cin,cout,shape", [("s2", 8, 16, (8, 12, 40)), ("s2", 16, 32, (6, 10, 34)), ("s2", 32, 64, (4, 8, 18)),
                                                     ("s1", 16, 16, (5, 7, 20)), ("deconv", 16, 8, (3, 5, 9))])
@pytest.mark.parametrize("poison", [False, True])
def test_conv3d_absmax_by_product_equals_a_pass_over_the_output(mode_name, cin, cout, shape, poison, dev):
    """mvd_conv3d_bn_relu_absmax_f32: max |y| over the FINITE outputs, bit-exact (a max, no arithmetic): fused into the store
    epilogue of the stride-2 kernel, a separate pass after the others; the output itself is the plain entry point's."""
    from robustmvd_amd import ops, _lib as L
    mode = {"s1": L.CONV3D_STRIDE1, "s2": L.CONV3D_STRIDE2, "deconv": L.DECONV3D_STRIDE2}[mode_name]
    g = torch.Generator().manual_seed(cin * 100 + cout)
    x = (torch.randn(2, *shape, cin, generator=g) * 3).to(dev)
    if poison:
        x[0, 1, 2, 3, 0] = float("inf")
        x[1, 0, 1, 1, 1] = float("nan")
    wshape = (cin, cout, 3, 3, 3) if mode_name == "deconv" else (cout, cin, 3, 3, 3)
    wt = (torch.randn(*wshape, generator=g) * 0.1).to(dev)
    sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), (torch.randn(cout, generator=g) * 0.1).to(dev)
    packed, _, _ = ops.pack_conv3d_weights(wt, mode)
    for relu in (True, False):
        want = ops.conv3d_bn_relu(x, packed, cin, cout, sc, sh, mode, relu=relu)
        got, amax = ops.conv3d_bn_relu(x, packed, cin, cout, sc, sh, mode, relu=relu, return_absmax=True)
        assert torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(want, nan=-7.0))
        assert float(amax) == _finite_absmax(want) > 0
        assert float(ops.absmax(want)) == _finite_absmax(want)


def test_warp_variance_absmax_by_product_equals_a_pass_over_the_volume(dev):
    from robustmvd_amd import ops
    from test_hip_shapes import mvs_inputs
    for (B, h, w, D, V, C, cl) in [(1, 24, 40, 8, 3, 32, True), (2, 19, 33, 5, 2, 32, True), (1, 16, 24, 4, 2, 8, False)]:
        feats, projs, key_inv, depth = mvs_inputs(B, C, h, w, D, V, seed=h + V)
        fs = [T(f, dev) for f in feats]
        vol, amax = ops.warp_variance(fs[0], fs[1:], [T(p, dev) for p in projs], T(key_inv, dev), T(depth, dev), channels_last=cl,
                                      return_absmax=True)
        assert float(amax) == _finite_absmax(vol) > 0
