"""Input resize, CPU side.  skimage — what the reference's ResizeInputs calls
(rmvd/data/transforms.py:64-66) — is not installed, so the IMAGE parity of this row is pinned one level down: against
scipy.ndimage.zoom(order=1, mode='mirror', grid_mode=True), the function skimage.transform.resize delegates to when no
axis shrinks (no anti-aliasing, float32 kept, clip a no-op).  That delegation is restated from skimage's published source:
"parity unpinned" for the wrapper, bit-exact for the arithmetic.  The intrinsics scaling is the reference's own formula."""
import numpy as np
import pytest
import scipy.ndimage as ndi

from oracle import mvd_oracle as O


@pytest.mark.parametrize("H,W,ht,wd", [(720, 1280, 768, 1280), (45, 70, 64, 128), (30, 50, 32, 64), (64, 64, 64, 128),
                                       (5, 7, 64, 64), (375, 1242, 384, 1280)])
def test_oracle_resize_is_scipy_zoom_bit_for_bit(H, W, ht, wd):
    rng = np.random.default_rng(H + W)
    img = rng.uniform(0, 255, (3, H, W)).astype(np.float32)
    ref = ndi.zoom(img, [1, 1 / (H / ht), 1 / (W / wd)], order=1, mode="mirror", cval=0, grid_mode=True)  # skimage's call
    assert ref.shape == (3, ht, wd) and ref.dtype == np.float32
    assert np.array_equal(O.resize_order1(img, ht, wd), ref)
    batched = np.stack([img, img[::-1]])
    assert np.array_equal(O.resize_order1(batched, ht, wd)[0], ref)


def test_oracle_resize_properties():
    # identity when the size is unchanged; a linear ramp is reproduced exactly in the interior (half-pixel centres)
    rng = np.random.default_rng(1)
    img = rng.uniform(0, 255, (3, 32, 64)).astype(np.float32)
    assert np.array_equal(O.resize_order1(img, 32, 64), img)
    H, W, ht, wd = 30, 50, 60, 100
    ramp = (np.arange(W, dtype=np.float32)[None, :] * 2.0 + np.arange(H, dtype=np.float32)[:, None] * 3.0)[None]
    out = O.resize_order1(ramp, ht, wd)[0]
    ys = (np.arange(ht) + 0.5) * H / ht - 0.5
    xs = (np.arange(wd) + 0.5) * W / wd - 0.5
    want = xs[None, :] * 2.0 + ys[:, None] * 3.0
    np.testing.assert_allclose(out[1:-1, 1:-1], want[1:-1, 1:-1], rtol=0, atol=1e-4)
    # mirror boundary (not clamp): the first output column of a 2x upscaling sits at x = -0.25 -> 0.75 p0 + 0.25 p1
    np.testing.assert_allclose(out[1:-1, 0], 0.75 * want[1:-1, 1] * 0 + (0.25 * 2.0 + ys[1:-1] * 3.0), atol=1e-4)
    with pytest.raises(ValueError):
        O.resize_order1(img, 16, 64)


def test_resize_inputs_scales_intrinsics_like_the_reference():
    K = np.array([[1000.0, 0, 640.0], [0, 1000.0, 360.0], [0, 0, 1]], np.float32)
    images, intr = O.resize_inputs([np.zeros((3, 720, 1280), np.float32)], [K], 768, 1280)
    assert images[0].shape == (3, 768, 1280) and intr[0].dtype == np.float32
    np.testing.assert_array_equal(intr[0], K * np.array([[1.0] * 3, [768 / 720] * 3, [1.0] * 3], np.float32))
