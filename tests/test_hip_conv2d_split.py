"""GPU parity of the split-operand 2-D convolution engine (mvd_conv2d_split_f32; Path A's DispNet blocks:
/root/reference rmvd/models/blocks/dispnet_encoder.py:6-27, dispnet_costvolume_encoder.py:7-50, dispnet_decoder.py:36-138,
learned_fusion.py:8-20).  Each layer kind against torch's float64 convolution of the same fp32 inputs on the CPU: the bar is
fp32-grade, i.e. no worse than what an fp32 convolution itself leaves (tolerance 3e-6 of the output scale, written below; an
fp32 accumulation of K terms is itself off by ~1e-6 .. 1e-5 of that scale)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _reference(x_nchw, wt, bias, stride, mode, act, slope):
    x64, w64 = x_nchw.double(), wt.double()
    b64 = None if bias is None else bias.double()
    if mode == 1:
        y = F.conv_transpose2d(x64, w64, b64, stride=2, padding=1)
    else:
        y = F.conv2d(x64, w64, b64, stride=stride, padding=wt.shape[-1] // 2)
    if act == 1:
        y = F.leaky_relu(y, slope)
    elif act == 2:
        y = F.relu(y)
    return y


CASES = [
    # (k, stride, mode, cin, cout, B, H, W, act)
    (3, 1, 0, 32, 64, 1, 20, 37, 1),      # 3x3 stride 1, 32-channel chunks
    (3, 1, 0, 64, 128, 2, 9, 18, 1),
    (3, 1, 0, 26, 32, 1, 17, 16, 1),      # Cin + 2 not a multiple of 8: padded slice, 8-channel chunks
    (3, 1, 0, 98, 32, 1, 33, 21, 1),      # rfeat5's channel count
    (3, 1, 0, 32, 2, 1, 19, 30, 0),       # a prediction head: 2 output channels, no activation
    (3, 1, 0, 128, 1, 2, 12, 18, 0),      # (the fusion block's 1x1 has 128 -> 1; here 3x3)
    (3, 2, 0, 16, 32, 1, 32, 48, 1),      # 3x3 stride 2
    (3, 2, 0, 64, 256, 2, 14, 22, 1),
    (5, 2, 0, 8, 16, 1, 40, 36, 1),       # 5x5 stride 2
    (5, 2, 0, 64, 128, 1, 30, 34, 1),
    (1, 1, 0, 64, 32, 1, 13, 29, 1),      # 1x1
    (1, 1, 0, 128, 1, 2, 12, 18, 0),
    (4, 2, 1, 32, 16, 1, 9, 14, 1),       # transposed 4x4 stride 2
    (4, 2, 1, 64, 32, 2, 16, 16, 1),
    (7, 2, 2, 3, 64, 2, 38, 52, 1),       # the first layer on the planar image
    (3, 1, 0, 512, 256, 1, 6, 9, 1),      # few pixels, many weights: the reduction is split over workgroups
    (3, 2, 0, 256, 512, 1, 12, 18, 1),
    (4, 2, 1, 512, 256, 1, 3, 5, 2),
    (3, 1, 0, 64, 64, 1, 16, 16, 2),      # ReLU
]


@pytest.mark.parametrize("k,stride,mode,cin,cout,B,H,W,act", CASES)
def test_conv2d_split_vs_float64(k, stride, mode, cin, cout, B, H, W, act, dev):
    from robustmvd_amd import ops
    g = torch.Generator().manual_seed(k * 1000 + cin + cout + H)
    x = torch.randn(B, cin, H, W, generator=g) * torch.rand(1, cin, 1, 1, generator=g) * 4
    wshape = (cin, cout, 4, 4) if mode == 1 else (cout, cin, k, k)
    wt = torch.randn(*wshape, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    slope = 0.2
    want = _reference(x, wt, bias, stride, mode, act, slope)
    wts = ops.pack_conv2d_weights_split(wt.to(dev), bias.to(dev), stride=stride, mode=mode)
    assert wts.cin_pad % 8 == 0 and wts.cin_pad >= cin
    if mode == 2:
        xin = x.to(dev)
    else:  # NHWC slice of a wider buffer: channels [4, 4 + cin_pad) of cin_pad + 12, pad channels zero
        buf = torch.full((B, H, W, wts.cin_pad + 12), 7.0, device=dev)
        buf[..., 4:4 + wts.cin_pad] = 0
        buf[..., 4:4 + cin] = x.permute(0, 2, 3, 1).to(dev)
        xin = buf[..., 4:4 + wts.cin_pad]
    Ho, Wo = want.shape[-2:]
    obuf = torch.full((B, Ho, Wo, cout + 8 if cout >= 4 else cout), -3.0, device=dev)
    out = obuf[..., 4:4 + cout] if cout >= 4 else obuf
    amax = torch.zeros(1, device=dev)
    got = ops.conv2d_split(xin, ops.absmax(x.to(dev)), wts, act=act, slope=slope, out=out, out_absmax=amax)
    assert got.data_ptr() == out.data_ptr()
    y = got.permute(0, 3, 1, 2).double().cpu()
    scale = float(want.abs().max())
    err = float((y - want).abs().max())
    assert err <= 3e-6 * scale, (err, scale)
    assert float(amax) == float(got.abs().max())
    if cout >= 4:  # the neighbouring channels of the destination buffer are untouched
        assert float(obuf[..., :4].min()) == -3.0 and float(obuf[..., 4 + cout:].max()) == -3.0
    # allocated output, no workspace (no split of the reduction): same values up to fp32 summation order
    got2 = ops.conv2d_split(xin, ops.absmax(x.to(dev)), wts, act=act, slope=slope, use_workspace=False)
    assert tuple(got2.shape) == (B, Ho, Wo, cout)
    assert float((got2.permute(0, 3, 1, 2).double().cpu() - want).abs().max()) <= 3e-6 * scale


@pytest.mark.parametrize("mag", [1e-30, 1e-8, 1.0, 1e4, 1e20])
def test_conv2d_split_range(mag, dev):
    """Block scaling: the same relative accuracy whatever the magnitude of the activations and of the weights."""
    from robustmvd_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 32, 12, 20, generator=g) * mag
    wt = torch.randn(16, 32, 3, 3, generator=g) * (1e-3 / mag if mag > 1 else 1e3)
    want = F.conv2d(x.double(), wt.double(), padding=1)
    wts = ops.pack_conv2d_weights_split(wt.to(dev))
    got = ops.conv2d_split(x.permute(0, 2, 3, 1).contiguous().to(dev), ops.absmax(x.to(dev)), wts, act=0)
    err = float((got.permute(0, 3, 1, 2).double().cpu() - want).abs().max())
    assert err <= 3e-6 * float(want.abs().max())


def test_conv2d_split_is_deterministic_and_refuses_bad_arguments(dev):
    from robustmvd_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 6, 9, 512, generator=g).to(dev)
    wt = (torch.randn(256, 512, 3, 3, generator=g) * 0.02).to(dev)
    wts = ops.pack_conv2d_weights_split(wt)
    am = ops.absmax(x)
    a = ops.conv2d_split(x, am, wts)
    b = ops.conv2d_split(x, am, wts)
    assert torch.equal(a, b)  # the split reduction adds its partial sums in a fixed order
    with pytest.raises(ValueError, match="not built"):
        ops.pack_conv2d_weights_split(torch.zeros(8, 8, 2, 2, device=dev))
    with pytest.raises(ValueError, match="channels"):
        ops.conv2d_split(x[..., :64], am, wts)
    with pytest.raises(ValueError, match="slice"):
        ops.conv2d_split(x.permute(0, 2, 1, 3), am, wts)


def test_upsample2x_into_matches_torch_interpolate(dev):
    from robustmvd_amd import ops
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 2, 7, 11, generator=g).to(dev)
    buf = torch.full((2, 14, 22, 12), 5.0, device=dev)
    am = torch.zeros(1, device=dev)
    ops.upsample2x_into(x, buf[..., 8:10], out_absmax=am)
    want = F.interpolate(x, size=(14, 22), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    # torch's formula and operation order; its GPU kernel contracts the blends into FMAs, this library does not: last-bit differences
    torch.testing.assert_close(buf[..., 8:10], want, atol=1e-6, rtol=1e-6)
    want_cpu = F.interpolate(x.cpu(), size=(14, 22), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    torch.testing.assert_close(buf[..., 8:10].cpu(), want_cpu, atol=1e-6, rtol=1e-6)
    assert float(am) == float(buf[..., 8:10].abs().max())
    assert float(buf[..., :8].min()) == 5.0 and float(buf[..., 10:].min()) == 5.0


@pytest.mark.parametrize("V", [2, 4])
def test_sweep_and_fusion_on_channel_last_layouts_equal_the_planar_entry_points(V, dev):
    """mvd_sweep_corr_nhwc_f32 / mvd_fuse_views_nhwc_f32: the same kernels' arithmetic on the layouts the 2-D engine works in ->
    bit-identical values, transposed."""
    from robustmvd_amd import ops
    import gen_common as gc
    from oracle import mvd_oracle as O
    N, C, h, w, S = 2, 64, 12, 20, 32
    rng = np.random.default_rng(V)
    fk = rng.standard_normal((N, C, h, w)).astype(np.float32)
    srcs = [rng.standard_normal((N, C, h, w)).astype(np.float32) for _ in range(V)]
    Kk = np.stack([np.array([[0.72, 0, 0.5], [0, 1.28, 0.5], [0, 0, 1]], np.float32)] * N)
    Ks = [Kk * np.array([[1.0 + 0.05 * v], [1.0 - 0.03 * v], [1.0]], np.float32) for v in range(V)]
    Ts = [np.stack([gc.synthetic_pose(rng, 0.08, 0.2) for _ in range(N)]) for _ in range(V)]
    inv = O.compute_sampling_invdepths(np.full(1, 0.4, np.float32), np.full(1, 100.0, np.float32), S)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    fk_t, srcs_t = t(fk), [t(s_) for s_ in srcs]
    args = (t(Kk), [t(k) for k in Ks], [t(x) for x in Ts], t(inv))
    corrs, masks = ops.sweep_corr(fk_t, srcs_t, *args)
    key_cl = fk_t.permute(0, 2, 3, 1).contiguous()
    bordered = []
    for s_ in srcs_t:
        b = torch.zeros(N, h + 3, w + 3, C, device=dev)
        b[:, 1:h + 1, 1:w + 1] = s_.permute(0, 2, 3, 1)
        bordered.append(b)
    big_c, big_m = torch.zeros(V, N, h, w, S + 8, device=dev), torch.zeros(V, N, h, w, S + 8, device=dev)
    c2 = [big_c[v, ..., 8:] for v in range(V)]
    m2 = [big_m[v, ..., 8:] for v in range(V)]
    cam = torch.zeros(1, device=dev)
    ops.sweep_corr_nhwc(key_cl, bordered, *args, c2, m2, corr_absmax=cam)
    assert float(cam) == max(float(c.abs().max()) for c in corrs)
    for v in range(V):
        assert torch.equal(c2[v].permute(0, 3, 1, 2), corrs[v]) and torch.equal(m2[v].permute(0, 3, 1, 2), masks[v])
    scores = [torch.randn(N, 1, h, w, device=dev) for _ in range(V)]
    fused, _ = ops.fuse_views(corrs, masks, scores)
    out = torch.full((N, h, w, S + 4), 9.0, device=dev)
    am = torch.zeros(1, device=dev)
    ops.fuse_views_nhwc(c2, m2, scores, out[..., 4:], out_absmax=am)
    assert torch.equal(out[..., 4:].permute(0, 3, 1, 2), fused)
    assert float(am) == float(fused.abs().max()) and float(out[..., :4].min()) == 9.0


# ---- K4's 3-D layers on the same kernel -------------------------------------------------------------------------------------
C3_CASES = [
    # (mode, cin, cout, B, D, h, w)   mode: 0 stride 1, 1 stride 2, 2 transposed
    (1, 8, 16, 1, 8, 20, 36), (1, 16, 32, 2, 6, 10, 18), (1, 32, 64, 1, 4, 8, 34),
    (0, 64, 64, 1, 5, 9, 20), (0, 32, 32, 1, 3, 17, 16), (0, 8, 8, 1, 4, 6, 40),
    (2, 64, 32, 1, 3, 5, 9), (2, 32, 16, 2, 4, 6, 18), (2, 16, 8, 1, 5, 12, 33),
]


@pytest.mark.parametrize("mode,cin,cout,B,D,h,w", C3_CASES)
@pytest.mark.parametrize("with_skip", [False, True])
def test_conv3d_igemm_vs_float64_and_fp32_kernel(mode, cin, cout, B, D, h, w, with_skip, dev):
    """mvd_conv3d_bn_relu_igemm_f32 (Conv3d 3x3x3 stride 1 / 2, ConvTranspose3d stride 2 with output_padding 1, as in
    mvsnet_components.py:78-101) against torch's float64 convolution on the CPU and against the fp32-MFMA kernel of the library."""
    from robustmvd_amd import ops, _lib as L
    g = torch.Generator().manual_seed(mode * 100 + cin + cout)
    x = torch.randn(B, cin, D, h, w, generator=g) * torch.rand(1, cin, 1, 1, 1, generator=g) * 3
    wshape = (cin, cout, 3, 3, 3) if mode == 2 else (cout, cin, 3, 3, 3)
    wt = torch.randn(*wshape, generator=g) * (2.0 / (cin * 27)) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    x64, w64 = x.double(), wt.double()
    if mode == 2:
        ref = F.conv_transpose3d(x64, w64, stride=2, padding=1, output_padding=1)
    else:
        ref = F.conv3d(x64, w64, stride=1 + mode, padding=1)
    ref = F.relu(ref * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1))
    skip = torch.randn(ref.shape, generator=g) if with_skip else None
    if with_skip:
        ref = ref + skip.double()
    xc = x.permute(0, 2, 3, 4, 1).contiguous().to(dev)
    sk = skip.permute(0, 2, 3, 4, 1).contiguous().to(dev) if with_skip else None
    packed = ops.pack_conv3d_weights_igemm(wt.to(dev), mode)
    got, amax = ops.conv3d_bn_relu_igemm(xc, ops.absmax(xc), packed, cin, cout, scale.to(dev), shift.to(dev), mode, relu=True, skip=sk,
                                         return_absmax=True)
    y = got.permute(0, 4, 1, 2, 3).double().cpu()
    sc = float(ref.abs().max())
    assert float((y - ref).abs().max()) <= 3e-6 * sc
    assert float(amax) == float(got.abs().max())
    w32, _, _ = ops.pack_conv3d_weights(wt.to(dev), mode)
    old = ops.conv3d_bn_relu(xc, w32, cin, cout, scale.to(dev), shift.to(dev), mode, relu=True, skip=sk)
    assert float((old - got).abs().max()) <= 2e-5 * sc


def test_conv3d_split_absmax_by_product(dev):
    from robustmvd_amd import ops
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(1, 9, 11, 37, 32, generator=g) * 2).to(dev)
    wt = (torch.randn(8, 32, 3, 3, 3, generator=g) * 0.05).to(dev)
    sc, sh = (torch.rand(8, generator=g) + 0.5).to(dev), (torch.randn(8, generator=g) * 0.1).to(dev)
    pk = ops.pack_conv3d_weights_split(wt)
    for relu in (True, False):
        want = ops.conv3d_bn_relu_split(x, pk, sc, sh, relu=relu)
        got, amax = ops.conv3d_bn_relu_split(x, pk, sc, sh, relu=relu, return_absmax=True)
        assert torch.equal(got, want) and float(amax) == float(want.abs().max()) > 0


@pytest.mark.parametrize("shape", [(2, 64, 128), (1, 37, 75), (3, 8, 64), (1, 5, 3)])
def test_conv2d_head_matches_float64_and_two_launches(shape, dev):
    """FeatureNet's conv0 -> conv1 as one launch (mvd_conv2d_head_f32; mvsnet_components.py:47-48) against the float64
    convolutions and against the two launches of the fp32-MFMA kernel: ragged tiles, images smaller than a tile, batch"""
    import torch.nn.functional as F
    from robustmvd_amd import ops
    B, H, W = shape
    g = torch.Generator(device="cpu").manual_seed(11)
    img = torch.randn(B, 3, H, W, generator=g)
    w0, w1 = torch.randn(8, 3, 3, 3, generator=g) * 0.3, torch.randn(8, 8, 3, 3, generator=g) * 0.2
    s0, b0, s1, b1 = (torch.randn(8, generator=g) for _ in range(4))
    ref = F.relu(F.conv2d(img.double(), w0.double(), padding=1) * s0.double().view(1, -1, 1, 1) + b0.double().view(1, -1, 1, 1))
    ref = F.relu(F.conv2d(ref, w1.double(), padding=1) * s1.double().view(1, -1, 1, 1) + b1.double().view(1, -1, 1, 1))
    d = lambda t: t.to(dev)
    args = (d(img), d(w0.permute(2, 3, 1, 0).contiguous()), d(s0), d(b0), d(w1.permute(2, 3, 1, 0).contiguous()), d(s1), d(b1))
    got = ops.conv2d_head(*args)
    assert tuple(got.shape) == (B, H, W, 8)
    got2, amax = ops.conv2d_head(*args, return_absmax=True)  # per-tile maxima, reduced: exactly max |y|
    assert torch.equal(got2, got) and float(amax) == float(got.abs().max())
    want = ref.permute(0, 2, 3, 1)
    scale = float(want.abs().max())
    assert float((got.double().cpu() - want).abs().max()) <= 2e-6 * scale
    p0, _, _, _ = ops.pack_conv2d_weights(d(w0))
    p1, _, _, _ = ops.pack_conv2d_weights(d(w1))
    two = ops.conv2d_bn_relu(ops.conv2d_bn_relu(d(img), p0, 3, 8, 3, 1, d(s0), d(b0)), p1, 8, 8, 3, 1, d(s1), d(b1))
    assert float((got - two).abs().max()) <= 2e-6 * scale
