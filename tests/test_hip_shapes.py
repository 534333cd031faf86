"""GPU parity on shapes the golden vectors do not cover — ragged sizes (not multiples of any tile), batch > 1,
other channel counts, every conv3d mode, V up to 6 — against the CPU oracle on identical seeded inputs, and
size-independent properties at the full BASELINE shapes (where the oracle would take too long)."""
import numpy as np
import pytest
import torch

import gen_common as gc
from oracle import c_oracle as CO
from oracle import mvd_oracle as O

pytestmark = pytest.mark.gpu
ATOL = RTOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def mvs_inputs(B, C, h, w, D, V, seed, rot=0.05, trans=0.15, dmin=0.5, dmax=10.0):
    rng = np.random.default_rng(seed)
    feats = [rng.standard_normal((B, C, h, w)).astype(np.float32) for _ in range(V + 1)]
    K = gc.synthetic_intrinsics(h * 4, w * 4)
    Ks = K.copy()
    Ks[:2] *= 0.25

    def proj(Tm, key):
        P = Tm.copy()
        P[:3, :4] = Ks @ P[:3, :4]
        return (np.linalg.inv(P) if key else P).astype(np.float32)

    key_inv = np.stack([proj(np.eye(4, dtype=np.float32), True)] * B)
    projs = [np.stack([proj(gc.synthetic_pose(rng, rot, trans), False) for _ in range(B)]) for _ in range(V)]
    depth = np.stack([np.linspace(dmin, dmax * (1 + 0.1 * b), D, dtype=np.float32) for b in range(B)])
    return feats, projs, key_inv, depth


@pytest.mark.parametrize("B,C,h,w,D,V", [(2, 32, 37, 53, 5, 3), (1, 32, 9, 70, 3, 6), (1, 8, 20, 33, 4, 2),
                                         (1, 64, 11, 17, 3, 1), (1, 32, 64, 96, 9, 4),
                                         # KITTI's 20 source views (rmvd/data/README.md:322-323), 12 and 16, the API's maximum
                                         (1, 32, 21, 30, 11, 20), (1, 32, 13, 22, 17, 12), (2, 32, 10, 19, 9, 16), (1, 32, 8, 12, 20, 32),
                                         (1, 16, 9, 14, 5, 20)])
@pytest.mark.parametrize("channels_last", [False, True])
def test_warp_variance_vs_oracle(B, C, h, w, D, V, channels_last, dev):
    from robustmvd_amd import ops
    feats, projs, key_inv, depth = mvs_inputs(B, C, h, w, D, V, seed=B * 1000 + h)
    ref = CO.warp_variance(feats[0], feats[1:], projs, key_inv, depth)
    got = ops.warp_variance(T(feats[0], dev), [T(f, dev) for f in feats[1:]], [T(p, dev) for p in projs], T(key_inv, dev),
                            T(depth, dev), channels_last=channels_last)
    if channels_last:
        got = got.permute(0, 4, 1, 2, 3)
    np.testing.assert_allclose(got.cpu().numpy(), ref, atol=ATOL, rtol=RTOL)


def test_warp_variance_wide_baseline_vs_oracle(dev):
    """large rotations / translations: most samples leave the image, some planes go behind a source camera"""
    from robustmvd_amd import ops
    feats, projs, key_inv, depth = mvs_inputs(1, 32, 24, 40, 8, 3, seed=77, rot=0.7, trans=0.8, dmin=0.2, dmax=3.0)
    ref = CO.warp_variance(feats[0], feats[1:], projs, key_inv, depth)
    got = ops.warp_variance(T(feats[0], dev), [T(f, dev) for f in feats[1:]], [T(p, dev) for p in projs], T(key_inv, dev),
                            T(depth, dev))
    np.testing.assert_allclose(got.cpu().numpy(), ref, atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("N,C,h,w,hs,ws,S,V,batched", [(2, 64, 13, 21, 11, 19, 7, 3, True), (1, 128, 5, 40, 5, 40, 9, 1, False),
                                                       (1, 256, 17, 16, 17, 16, 33, 2, False),
                                                       (1, 64, 9, 14, 9, 14, 5, 20, False), (2, 64, 6, 9, 7, 8, 4, 32, True)])
def test_sweep_corr_vs_oracle(N, C, h, w, hs, ws, S, V, batched, dev):
    from robustmvd_amd import ops
    rng = np.random.default_rng(N * 100 + C)
    fk = rng.standard_normal((N, C, h, w)).astype(np.float32)
    fs = [rng.standard_normal((N, C, hs, ws)).astype(np.float32) for _ in range(V)]
    Kk = np.stack([np.array([[0.72, 0, 0.5], [0, 1.28, 0.5], [0, 0, 1]], np.float32)] * N)
    Ks = [Kk * np.array([[1.0 + 0.05 * v], [1.0 - 0.03 * v], [1.0]], np.float32) for v in range(V)]
    Ts = [np.stack([gc.synthetic_pose(rng, 0.08, 0.2) for _ in range(N)]) for _ in range(V)]
    inv = O.compute_sampling_invdepths(np.full(N if batched else 1, 0.4, np.float32),
                                       np.linspace(100.0, 1000.0, N if batched else 1).astype(np.float32), S)
    ref_c, ref_m = CO.sweep_corr(fk, fs, Kk, Ks, Ts, inv)
    corrs, masks = ops.sweep_corr(T(fk, dev), [T(f, dev) for f in fs], T(Kk, dev), [T(k, dev) for k in Ks],
                                  [T(t, dev) for t in Ts], T(inv, dev))
    for v in range(V):
        m = masks[v].cpu().numpy()
        mism = m != ref_m[v]
        assert mism.mean() <= 1e-4
        np.testing.assert_allclose(corrs[v].cpu().numpy()[~mism], ref_c[v][~mism], atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("V", [2, 3, 5, 20, 32])
def test_fuse_views_vs_oracle(V, dev):
    from robustmvd_amd import ops
    rng = np.random.default_rng(V)
    N, S, h, w = 2, 19, 7, 11
    corrs = [rng.standard_normal((N, S, h, w)).astype(np.float32) for _ in range(V)]
    masks = [(rng.uniform(size=(N, S, h, w)) > 0.4).astype(np.float32) for _ in range(V)]
    for m in masks:
        m[:, 3:6, 2:4, 5:] = 0
    scores = [rng.standard_normal((N, 1, h, w)).astype(np.float32) * 3 for _ in range(V)]
    rf, rm = CO.fuse_views(corrs, masks, scores)
    f, m = ops.fuse_views([T(c, dev) for c in corrs], [T(x, dev) for x in masks], [T(s, dev) for s in scores])
    assert (m.cpu().numpy() == rm).all()
    np.testing.assert_allclose(f.cpu().numpy(), rf, atol=1e-5, rtol=1e-5)


CONV_CASES = [  # (Cin, Cout, mode, D, h, w, relu, skip)
    (32, 8, 0, 5, 7, 50, True, False), (32, 8, 0, 18, 9, 70, True, False), (8, 1, 0, 5, 6, 70, False, False),
    (8, 1, 0, 19, 5, 9, False, True), (16, 16, 0, 3, 5, 19, True, False), (32, 32, 0, 4, 6, 35, True, True),
    (64, 64, 0, 3, 5, 17, True, False), (8, 16, 1, 6, 10, 38, True, False), (16, 32, 1, 4, 8, 66, True, False),
    (32, 64, 1, 4, 6, 34, True, False), (64, 32, 2, 2, 3, 9, True, True), (32, 16, 2, 3, 5, 20, True, True),
    (16, 8, 2, 3, 6, 70, True, True), (8, 8, 0, 4, 5, 33, False, False), (16, 8, 0, 4, 5, 33, True, False),
    (32, 8, 2, 2, 5, 33, False, False), (64, 8, 2, 2, 3, 17, True, True), (8, 8, 2, 3, 4, 65, True, False),
    # widths that select the other tile widths (best_mt): 64-wide generic / pair tiles, 32-wide pair, 16-wide stride-2
    (16, 16, 0, 3, 5, 64, True, False), (8, 16, 1, 4, 8, 128, True, False), (16, 8, 2, 2, 4, 64, True, True),
    (16, 8, 2, 2, 4, 32, True, False), (32, 64, 1, 4, 6, 96, True, False), (32, 64, 1, 4, 6, 64, True, False),
]


@pytest.mark.parametrize("Cin,Cout,mode,D,h,w,relu,skip", CONV_CASES)
def test_conv3d_modes_vs_oracle(Cin, Cout, mode, D, h, w, relu, skip, dev):
    """every conv3d mode (incl. the PAIR and depth-marching forms, which need >= 1024 workgroups to be picked:
    the second conv0 case is sized for that) on ragged extents, against the C oracle"""
    from robustmvd_amd import ops
    rng = np.random.default_rng(Cin * 100 + Cout + mode)
    x = rng.standard_normal((Cin, D, h, w)).astype(np.float32)
    wshape = (Cin, Cout, 3, 3, 3) if mode == 2 else (Cout, Cin, 3, 3, 3)
    wgt = (rng.standard_normal(wshape) * np.sqrt(2.0 / (27 * Cin))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = (rng.standard_normal(Cout) * 0.1).astype(np.float32)
    if mode == 2:
        sk = rng.standard_normal((Cout, 2 * D, 2 * h, 2 * w)).astype(np.float32) if skip else None
        ref = CO.deconv3d(x, wgt, scale, shift, relu=relu, skip=sk)
    else:
        s = 2 if mode == 1 else 1
        sk = rng.standard_normal((Cout, D // s, h // s, w // s)).astype(np.float32) if skip else None
        ref = CO.conv3d(x, wgt, scale, shift, stride=s, relu=relu, skip=sk)
    packed, cin, cout = ops.pack_conv3d_weights(T(wgt, dev), mode)
    assert (cin, cout) == (Cin, Cout)
    xt = T(x, dev)[None].permute(0, 2, 3, 4, 1).contiguous()
    skt = T(sk, dev)[None].permute(0, 2, 3, 4, 1).contiguous() if skip else None
    y = ops.conv3d_bn_relu(xt, packed, Cin, Cout, T(scale, dev), T(shift, dev), mode, relu=relu, skip=skt)
    np.testing.assert_allclose(y[0].permute(3, 0, 1, 2).cpu().numpy(), ref, atol=ATOL, rtol=RTOL)


def test_conv0_marching_large_grid_vs_oracle(dev):
    """Grids with >= 1024 depth-marching workgroups (8-plane chunks), so that the kernels the headline shape runs are the
    ones under test: conv0's k-split PAIR kernel (odd width, ragged rows, a partial last chunk), the 16 -> 16 marching
    kernel, and the prob kernel — against the C oracle."""
    from robustmvd_amd import ops
    rng = np.random.default_rng(5)
    B, Cin, Cout, D, h, w = 1, 32, 8, 180, 62, 71   # 3 x 16 x 23 = 1104 workgroups of 8 planes
    x = rng.standard_normal((B, Cin, D, h, w)).astype(np.float32)
    wgt = (rng.standard_normal((Cout, Cin, 3, 3, 3)) * 0.05).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = (rng.standard_normal(Cout) * 0.1).astype(np.float32)
    ref = np.stack([CO.conv3d(x[b], wgt, scale, shift) for b in range(B)])
    packed, _, _ = ops.pack_conv3d_weights(T(wgt, dev), 0)
    y = ops.conv3d_bn_relu(T(x, dev).permute(0, 2, 3, 4, 1).contiguous(), packed, Cin, Cout, T(scale, dev), T(shift, dev), 0)
    np.testing.assert_allclose(y.permute(0, 4, 1, 2, 3).cpu().numpy(), ref, atol=ATOL, rtol=RTOL)
    del x, ref, y
    # the same kernel with TWO batch elements (2 x 24 x 12 x 2 = 1152 workgroups of 8 planes; the batch index is the slowest
    # part of its block decode)
    B2, D2b, h2b, w2b = 2, 96, 96, 64
    xb = rng.standard_normal((B2, Cin, D2b, h2b, w2b)).astype(np.float32)
    refb = np.stack([CO.conv3d(xb[b], wgt, scale, shift) for b in range(B2)])
    yb = ops.conv3d_bn_relu(T(xb, dev).permute(0, 2, 3, 4, 1).contiguous(), packed, Cin, Cout, T(scale, dev), T(shift, dev), 0)
    np.testing.assert_allclose(yb.permute(0, 4, 1, 2, 3).cpu().numpy(), refb, atol=ATOL, rtol=RTOL)
    del xb, refb, yb
    # 16 -> 16 (conv2's form): 2 x 16 x 33 = 1056 workgroups of 8 planes
    D2, h2, w2 = 264, 64, 100
    x16 = rng.standard_normal((1, 16, D2, h2, w2)).astype(np.float32)
    w16 = (rng.standard_normal((16, 16, 3, 3, 3)) * 0.07).astype(np.float32)
    sc16 = rng.uniform(0.5, 1.5, 16).astype(np.float32)
    sh16 = (rng.standard_normal(16) * 0.1).astype(np.float32)
    ref16 = CO.conv3d(x16[0], w16, sc16, sh16)
    pk16, _, _ = ops.pack_conv3d_weights(T(w16, dev), 0)
    y16 = ops.conv3d_bn_relu(T(x16, dev).permute(0, 2, 3, 4, 1).contiguous(), pk16, 16, 16, T(sc16, dev), T(sh16, dev), 0)
    np.testing.assert_allclose(y16[0].permute(3, 0, 1, 2).cpu().numpy(), ref16, atol=ATOL, rtol=RTOL)
    del x16, ref16, y16
    B, D, h, w = 2, 40, 64, 96
    # prob layer, same grid
    wp = (rng.standard_normal((1, 8, 3, 3, 3)) * 0.1).astype(np.float32)
    x8 = rng.standard_normal((B, 8, D, h, w)).astype(np.float32)
    refp = np.stack([CO.conv3d(x8[b], wp, np.ones(1, np.float32), np.array([0.3], np.float32), relu=False) for b in range(B)])
    pk, _, _ = ops.pack_conv3d_weights(T(wp, dev), 0)
    yp = ops.conv3d_bn_relu(T(x8, dev).permute(0, 2, 3, 4, 1).contiguous(), pk, 8, 1, torch.ones(1, device=dev),
                            torch.full((1,), 0.3, device=dev), 0, relu=False)
    np.testing.assert_allclose(yp[..., 0].cpu().numpy(), refp[:, 0], atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("Cin,Cout,k,stride,B,h,w,relu,layout", [
    (3, 8, 3, 1, 2, 37, 70, True, "nhwc"), (3, 16, 5, 2, 1, 33, 45, True, "nchw"), (3, 32, 3, 1, 1, 9, 130, False, "nhwc"),
    (8, 8, 3, 1, 1, 40, 64, True, "nhwc"), (8, 16, 5, 2, 2, 38, 66, True, "nhwc"), (8, 32, 5, 2, 1, 21, 35, True, "border"),
    (16, 16, 3, 1, 1, 17, 65, True, "nchw"), (16, 32, 5, 2, 1, 30, 50, True, "nhwc"), (16, 8, 3, 1, 1, 8, 16, False, "border"),
    (32, 32, 3, 1, 2, 19, 33, True, "nhwc"), (32, 32, 3, 1, 1, 12, 40, False, "border"), (32, 16, 5, 2, 1, 23, 31, True, "nchw"),
    (32, 8, 3, 1, 1, 1, 1, True, "nhwc"),
])
def test_conv2d_layers_vs_oracle(Cin, Cout, k, stride, B, h, w, relu, layout, dev):
    """K6: every (Cin, Cout-tile, kernel) kernel family on shapes ragged against the tiles, all three output layouts."""
    from oracle import pipeline as P
    from robustmvd_amd import ops
    from robustmvd_amd import _lib as L
    rng = np.random.default_rng(Cin * 1000 + Cout * 10 + k)
    x = rng.standard_normal((B, Cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((Cout, Cin, k, k)) * np.sqrt(2.0 / (Cin * k * k))).astype(np.float32)
    bn = (rng.uniform(0.5, 1.5, Cout).astype(np.float32), (rng.standard_normal(Cout) * 0.1).astype(np.float32),
          (rng.standard_normal(Cout) * 0.1).astype(np.float32), rng.uniform(0.5, 1.5, Cout).astype(np.float32))
    want = P.conv_bn_relu_2d(x, wt, bn, stride=stride, relu=relu)
    scale = bn[0] / np.sqrt(bn[3] + np.float32(1e-5))
    shift = bn[1] - bn[2] * scale
    xt = torch.from_numpy(x).to(dev)
    packed, _, _, _ = ops.pack_conv2d_weights(torch.from_numpy(wt).to(dev))
    xin = xt if Cin == 3 else xt.permute(0, 2, 3, 1).contiguous()
    lay = {"nhwc": L.LAYOUT_NHWC, "nchw": L.LAYOUT_NCHW, "border": L.LAYOUT_NHWC_BORDER}[layout]
    y = ops.conv2d_bn_relu(xin, packed, Cin, Cout, k, stride, torch.from_numpy(scale).to(dev), torch.from_numpy(shift).to(dev),
                           relu=relu, out_layout=lay)
    ho, wo = want.shape[2:]
    if layout == "nhwc":
        got = y.permute(0, 3, 1, 2)
    elif layout == "border":
        assert tuple(y.shape) == (B, ho + 3, wo + 3, Cout)
        got = y[:, 1:ho + 1, 1:wo + 1].permute(0, 3, 1, 2)
        assert float(y.abs().sum()) == pytest.approx(float(got.abs().sum()), rel=1e-6)  # nothing written outside the interior
    else:
        got = y
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=1e-4, rtol=1e-4)


def test_warp_variance_staged_features_match_repacked(dev):
    """MVD_FEAT_NHWC_BORDER: handing K3 the zero-bordered channel-last maps gives bit-identical volumes."""
    from robustmvd_amd import ops
    B, C, h, w, D, V = 2, 32, 21, 38, 6, 3
    feats, projs, key_inv, depth = mvs_inputs(B, C, h, w, D, V, seed=77)
    ft = [T(f, dev) for f in feats]
    staged = []
    for f in ft:
        s_ = torch.zeros(B, h + 3, w + 3, C, device=dev)
        s_[:, 1:h + 1, 1:w + 1] = f.permute(0, 2, 3, 1)
        staged.append(s_)
    args = ([T(p, dev) for p in projs], T(key_inv, dev), T(depth, dev))
    for cl in (False, True):
        a = ops.warp_variance(ft[0], ft[1:], *args, channels_last=cl)
        b = ops.warp_variance(staged[0], staged[1:], *args, channels_last=cl, staged=True)
        assert torch.equal(a, b)


def test_conv2d_rejects_bad_arguments(dev):
    from robustmvd_amd import ops
    x = torch.zeros(1, 5, 5, 8, device=dev)
    with pytest.raises(ValueError):
        ops.pack_conv2d_weights(torch.zeros(8, 7, 3, 3, device=dev))
    packed, _, _, _ = ops.pack_conv2d_weights(torch.zeros(8, 8, 3, 3, device=dev))
    one = torch.ones(8, device=dev)
    with pytest.raises(ValueError):
        ops.conv2d_bn_relu(x, packed, 8, 8, 3, 2, one, one)
    with pytest.raises(ValueError):
        ops.conv2d_bn_relu(x.permute(0, 3, 1, 2).contiguous(), packed, 8, 8, 3, 1, one, one)


@pytest.mark.parametrize("B,D,h,w", [(2, 2, 3, 5), (1, 1, 4, 4), (1, 96, 33, 47)])
def test_softmax_regress_vs_oracle(B, D, h, w, dev):
    from robustmvd_amd import ops
    rng = np.random.default_rng(D)
    cost = (rng.standard_normal((B, D, h, w)) * 4).astype(np.float32)
    dv = np.stack([np.linspace(0.5, 10.0, D, dtype=np.float32)] * B) if D > 1 else np.full((B, 1), 2.5, np.float32)
    depth, conf = ops.softmax_regress(T(cost, dev), T(dv, dev))
    rd, rc, _ = O.softmax_regress(cost, dv)
    np.testing.assert_allclose(depth.cpu().numpy(), rd, atol=1e-5, rtol=1e-5)
    assert np.isclose(conf.cpu().numpy(), rc, atol=1e-5, rtol=1e-5).mean() > 0.999


# ---------------------------------------------------------------------------------------------
# full BASELINE shapes: properties that need no oracle
# ---------------------------------------------------------------------------------------------
def test_full_size_warp_variance_properties(dev):
    """headline shape 192x288x256 planes, V=4: (1) homo_warp is linear in the features, (2) the variance does not
    depend on the order of the source views, (3) var >= -eps, (4) both output layouts agree bit for bit, (5) a D-slab
    of the big launch equals a small launch on that slab (checked against the oracle on 2 planes)."""
    from robustmvd_amd import ops
    B, C, h, w, D, V = 1, 32, 192, 288, 256, 4
    feats, projs, key_inv, depth = mvs_inputs(B, C, h, w, D, V, seed=3)
    ft = [T(f, dev) for f in feats]
    pt = [T(p, dev) for p in projs]
    ki, dt = T(key_inv, dev), T(depth, dev)
    var = ops.warp_variance(ft[0], ft[1:], pt, ki, dt, channels_last=True)
    assert float(var.min()) > -1e-4
    perm = [2, 0, 3, 1]
    var_p = ops.warp_variance(ft[0], [ft[1 + i] for i in perm], [pt[i] for i in perm], ki, dt, channels_last=True)
    assert float((var - var_p).abs().max()) < 2e-5
    var_ncdhw = ops.warp_variance(ft[0], ft[1:], pt, ki, dt, channels_last=False)
    assert torch.equal(var_ncdhw.permute(0, 2, 3, 4, 1), var)
    del var_p, var_ncdhw
    sl = [17, 200]
    small = CO.warp_variance(feats[0], feats[1:], projs, key_inv, depth[:, sl])
    # default (folded, rcp) sampling positions are within 1e-4 px of the reference chain; on white-noise features
    # (unit slope per pixel) at coordinates up to 288 that shows as a ~1e-5 fraction of elements beyond 1e-4
    got = var[0, sl].permute(3, 0, 1, 2).cpu().numpy()
    bad = ~np.isclose(got, small[0], atol=ATOL, rtol=RTOL)
    assert bad.mean() < 2e-4
    np.testing.assert_allclose(got, small[0], atol=2e-3, rtol=2e-3)
    # MVD_GRID_EXACT follows the reference's operation chain rounding for rounding; what is left is the rounding of
    # the 4x4 projection product (an fmaf chain here, a plain sum in the oracle, BLAS in the reference): measured
    # 4e-6 of the elements beyond 1e-4 (fast grid: 4e-5), max 3e-4
    got_x = ops.warp_variance(ft[0], ft[1:], pt, ki, dt, channels_last=True, exact_grid=True)[0, sl].permute(3, 0, 1, 2).cpu().numpy()
    assert (~np.isclose(got_x, small[0], atol=ATOL, rtol=RTOL)).mean() < 2e-5
    np.testing.assert_allclose(got_x, small[0], atol=1e-3, rtol=1e-3)
    del var, got_x
    a, b = 0.7, -1.3
    w1 = ops.homo_warp(ft[1], pt[0], ki, dt[:, :32])
    w2 = ops.homo_warp(ft[2], pt[0], ki, dt[:, :32])
    w12 = ops.homo_warp(a * ft[1] + b * ft[2], pt[0], ki, dt[:, :32])
    assert float((w12 - (a * w1 + b * w2)).abs().max()) < 2e-5


def test_full_size_regulariser_and_regression_properties(dev):
    """headline volume through CostRegNet + soft argmin: finite, depth inside the sampled range, confidence in
    [0, 1], and a D-slab interior equals the small run of the same slab (receptive-field check vs the oracle)."""
    import robustmvd_amd as R
    from robustmvd_amd import ops
    from test_oracle_golden import costreg_shapes
    D, h, w = 256, 192, 288
    net = R.CostRegNet().eval()
    sd = gc.fill_state_dict(costreg_shapes(), 11)
    full = net.state_dict()
    for k, v in sd.items():
        full[k] = torch.from_numpy(v)
    net.load_state_dict(full)
    net = net.to(dev)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.rand((1, D, h, w, 32), device=dev, generator=g)
    cost = net.forward_channels_last(x)
    assert tuple(cost.shape) == (1, D, h, w) and bool(torch.isfinite(cost).all())
    dv = torch.linspace(0.5, 10.0, D, device=dev)[None]
    depth, conf = ops.softmax_regress(cost, dv)
    assert float(depth.min()) >= 0.5 - 1e-4 and float(depth.max()) <= 10.0 + 1e-4
    assert float(conf.min()) >= 0.0 and float(conf.max()) <= 1.0 + 1e-5
    # a 32 x 32 x 48 crop re-run on its own agrees away from the crop border (receptive field of the U-Net < 30)
    crop = x[:, 96:160, 64:128, 96:192].contiguous()
    cc = net.forward_channels_last(crop)
    ref = CO.cost_reg_net(crop[0].permute(3, 0, 1, 2).cpu().numpy()[None], sd)[0, 0]
    np.testing.assert_allclose(cc[0].cpu().numpy(), ref, atol=2e-4, rtol=1e-3)


@pytest.mark.parametrize("cfg", ["M4,2,4", "L4", "T8,128,8,2", "T8,96,2,1", "T16,128,3,2", "T16,104,8,1", "T32,128,1,2", "T32,104,32,1", "T8,104,16,1", "T8,128,16,1", "lds,4", "lds,8", "wave,4", "wave,8", "4,3", "8,2", "r4,3", "u4,2", "v4,3", "v4,4", "q8,4", "q8,3"])
def test_warp_variance_experimental_variants_match_default(cfg, dev, monkeypatch):
    """the experimental forms of K3 (LDS-staged footprints, other plane/occupancy splits; compiled only into
    robustmvd_amd/lib_exp/libmvd_hip_exp.so, never into the product library) give the same volume as the product
    kernel bit for bit, including tiles whose footprint falls back to direct gathers"""
    import os
    from robustmvd_amd import _lib as L
    from robustmvd_amd import ops
    if not os.path.exists(L.EXP_LIB_PATH):
        pytest.skip("the experiments library is not built (make -C robustmvd_amd/csrc exp)")
    feats, projs, key_inv, depth = mvs_inputs(1, 32, 45, 70, 19, 3, seed=9, rot=0.12, trans=0.3, dmin=0.4, dmax=8.0)
    args = (T(feats[0], dev), [T(f, dev) for f in feats[1:]], [T(p, dev) for p in projs], T(key_inv, dev), T(depth, dev))
    monkeypatch.setenv("MVD_K3_CFG", cfg)
    ref = ops.warp_variance(*args, channels_last=True)  # product library: ignores the environment
    with L.use_experiments_library():
        got = ops.warp_variance(*args, channels_last=True)
    assert torch.equal(got, ref)


def test_largest_baseline_shape_runs(dev):
    """BASELINE configs[4] (704x1280, 6 source views, 512 planes: a 3.7 GB variance volume, > 2^31 bytes per tensor):
    the whole Path-B hot path runs, stays finite, and a D-slab of K3 matches the oracle (64-bit offsets everywhere)."""
    import robustmvd_amd as R
    from robustmvd_amd import ops
    B, C, h, w, D, V = 1, 32, 176, 320, 512, 6
    feats, projs, key_inv, depth = mvs_inputs(B, C, h, w, D, V, seed=21)
    ft = [T(f, dev) for f in feats]
    pt = [T(p, dev) for p in projs]
    var = ops.warp_variance(ft[0], ft[1:], pt, T(key_inv, dev), T(depth, dev), channels_last=True)
    assert var.numel() * 4 > 2 ** 31
    sl = [3, 300, 511]
    small = CO.warp_variance(feats[0], feats[1:], projs, key_inv, depth[:, sl])
    got = var[0, sl].permute(3, 0, 1, 2).cpu().numpy()
    assert (~np.isclose(got, small[0], atol=ATOL, rtol=RTOL)).mean() < 2e-4
    np.testing.assert_allclose(got, small[0], atol=2e-3, rtol=2e-3)
    net = R.CostRegNet().eval().to(dev)
    cost = net.forward_channels_last(var)
    del var
    assert tuple(cost.shape) == (1, D, h, w) and bool(torch.isfinite(cost).all())
    depth_map, conf = ops.softmax_regress(cost, T(depth, dev))
    assert bool(torch.isfinite(depth_map).all()) and float(conf.max()) <= 1.0 + 1e-5


def test_mvsnet_20_source_views_vs_oracle_pipeline(dev):
    """KITTI's evaluation feeds 20 source views (rmvd/data/README.md:322-323): model.run with 21 images against the
    end-to-end oracle"""
    import robustmvd_amd as R
    from oracle import pipeline as PL
    H, W, D, V = 64, 96, 16, 20
    model = R.MVSNet(num_sampling_steps=D).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = gc.fill_state_dict(shapes, 32)
    full = model.state_dict()
    for k, v in sd.items():
        full[k] = torch.from_numpy(v)
    model.load_state_dict(full)
    model = R.add_run_function(model.to(dev))
    s = gc.synthetic_sample(43, H, W, V)
    pred, _ = model.run(images=s["images"], poses=s["poses"], intrinsics=s["intrinsics"], keyview_idx=0,
                        depth_range=(np.float32(0.5), np.float32(10.0)))
    assert pred["depth"].shape == (1, H // 4, W // 4)
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)
    ref = PL.mvsnet_forward([((im[None] / 255.0 - mean) / std).astype(np.float32) for im in s["images"]], [p[None] for p in s["poses"]],
                            [k[None] for k in s["intrinsics"]], 0, (0.5, 10.0), sd, D)
    np.testing.assert_allclose(pred["depth"], ref["depth"][0], rtol=1e-3)


def test_robustmvd_20_source_views_vs_oracle_pipeline(dev):
    """Path A with KITTI's 20 source views: sweep (K1) and learned fusion (K2) over 20 views, inverse-depth space"""
    import robustmvd_amd as R
    from oracle import pipeline as PL
    H, W, V = 128, 192, 20
    model = R.RobustMVD().eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = gc.robustmvd_weights(shapes, 6)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = R.add_run_function(model.to(dev))
    s = gc.synthetic_sample(44, H, W, V)
    pred, aux = model.run(images=s["images"], poses=s["poses"], intrinsics=s["intrinsics"], keyview_idx=0)
    scale = np.array([[W] * 3, [H] * 3, [1.0] * 3], np.float32)
    ref = PL.robustmvd_forward([(im / 255.0 - 0.4).astype(np.float32)[None] for im in s["images"]],
                               [p[None] for p in s["poses"]], [(k / scale)[None] for k in s["intrinsics"]], 0, sd)
    assert pred["depth"].shape == (1, H // 2, W // 2)
    np.testing.assert_allclose(aux["invdepth"], ref["invdepth"][0], atol=1e-4, rtol=1e-4)


def test_mvsnet_batch2_keyview1_vs_oracle_pipeline(dev):
    """model protocol with a batch of 2 and the key view in the middle of the list, against the end-to-end oracle"""
    import robustmvd_amd as R
    from oracle import pipeline as PL
    H, W, D, V = 64, 96, 16, 2
    model = R.MVSNet(num_sampling_steps=D).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = gc.fill_state_dict(shapes, 31)
    full = model.state_dict()
    for k, v in sd.items():
        full[k] = torch.from_numpy(v)
    model.load_state_dict(full)
    model = R.add_run_function(model.to(dev))
    s0, s1 = gc.synthetic_sample(40, H, W, V), gc.synthetic_sample(41, H, W, V)
    order = [1, 0, 2]  # key view second
    images = [np.stack([s0["images"][i], s1["images"][i]]) for i in order]
    poses = [np.stack([s0["poses"][i], s1["poses"][i]]) for i in order]
    intr = [np.stack([s0["intrinsics"][i], s1["intrinsics"][i]]) for i in order]
    key = np.array([1, 1])
    dr = (np.array([0.5, 0.5], np.float32), np.array([10.0, 10.0], np.float32))
    pred, _ = model.run(images=images, poses=poses, intrinsics=intr, keyview_idx=key, depth_range=dr)
    assert pred["depth"].shape == (2, 1, H // 4, W // 4)
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)
    ref = PL.mvsnet_forward([((im / 255.0 - mean) / std).astype(np.float32) for im in images], poses, intr, 1, (0.5, 10.0), sd, D)
    np.testing.assert_allclose(pred["depth"], ref["depth"], rtol=1e-3)
    np.testing.assert_allclose(pred["depth_uncertainty"], ref["depth_uncertainty"], atol=2e-3)
