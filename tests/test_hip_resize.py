"""GPU tests of the on-device input resize: the kernel against
scipy.ndimage.zoom — the function skimage.transform.resize (not installed: "parity unpinned" for that wrapper, see
tests/test_resize_cpu.py) delegates to — and the model adapters on inputs whose size is not a multiple of 64 / 32, e.g.
the 720x1280 layout of the reference's own sample_data (inference.py:18-55)."""
import numpy as np
import pytest
import scipy.ndimage as ndi
import torch

import gen_common as gc
from oracle import mvd_oracle as O
from oracle import pipeline as PL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("shape,ht,wd", [((3, 720, 1280), 768, 1280), ((2, 3, 45, 70), 64, 128), ((1, 375, 1242), 384, 1280),
                                         ((3, 64, 64), 64, 64), ((3, 5, 7), 64, 64)])
def test_resize_kernel_is_scipy_zoom_bit_for_bit(shape, ht, wd, dev):
    from robustmvd_amd import ops
    img = np.random.default_rng(ht + wd).uniform(0, 255, shape).astype(np.float32)
    H, W = shape[-2:]
    zoom = [1] * (len(shape) - 2) + [1 / (H / ht), 1 / (W / wd)]
    ref = ndi.zoom(img, zoom, order=1, mode="mirror", cval=0, grid_mode=True)
    got = ops.resize_order1(torch.from_numpy(img).to(dev), ht, wd).cpu().numpy()
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)
    assert np.array_equal(got, O.resize_order1(img, ht, wd))
    with pytest.raises(ValueError):
        ops.resize_order1(torch.from_numpy(img).to(dev), H - 1, wd)


def test_robustmvd_runs_on_sample_data_layout(dev):
    """720x1280 -> resized to 768x1280 on the device, intrinsics rescaled; output at half the RESIZED resolution like the
    reference (robust_mvd.py:104-120).  Compared with the oracle pipeline fed the oracle-resized inputs."""
    import robustmvd_amd as R
    H, W, V = 720, 1280, 1
    model = R.RobustMVD().eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = gc.robustmvd_weights(shapes, 5)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = R.add_run_function(model.to(dev))
    s = gc.synthetic_sample(11, H, W, V)
    ad = model.input_adapter(images=[im[None] for im in s["images"]], keyview_idx=np.array([0]),
                             poses=[p[None] for p in s["poses"]], intrinsics=[k[None] for k in s["intrinsics"]])
    ims, intr = O.resize_inputs(s["images"], s["intrinsics"], 768, 1280)
    scale = np.array([[1280] * 3, [768] * 3, [1.0] * 3], np.float32)
    for a, b in zip(ad["images"], ims):
        assert tuple(a.shape) == (1, 3, 768, 1280)
        assert np.array_equal(a[0].cpu().numpy(), (b / 255.0 - 0.4).astype(np.float32))
    for a, b in zip(ad["intrinsics"], intr):
        np.testing.assert_array_equal(a[0].cpu().numpy(), b / scale)
    pred, aux = model.run(images=s["images"], poses=s["poses"], intrinsics=s["intrinsics"], keyview_idx=0)
    assert pred["depth"].shape == (1, 384, 640)
    ref = PL.robustmvd_forward([(im / 255.0 - 0.4).astype(np.float32)[None] for im in ims], [p[None] for p in s["poses"]],
                               [(k / scale)[None] for k in intr], 0, sd)
    np.testing.assert_allclose(aux["invdepth"], ref["invdepth"][0], atol=1e-4, rtol=1e-4)


def test_mvsnet_runs_on_non_multiple_of_32(dev):
    """MVSNet adapter: UpscaleInputsToNextMultipleOf(32) (mvsnet.py:178) on the device, then the usual normalisation."""
    import robustmvd_amd as R
    H, W, V, D = 100, 150, 2, 16
    model = R.MVSNet(num_sampling_steps=D).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = gc.fill_state_dict(shapes, 13)
    full = model.state_dict()
    for k, v in sd.items():
        full[k] = torch.from_numpy(v)
    model.load_state_dict(full)
    model = R.add_run_function(model.to(dev))
    s = gc.synthetic_sample(12, H, W, V)
    pred, _ = model.run(images=s["images"], poses=s["poses"], intrinsics=s["intrinsics"], keyview_idx=0,
                        depth_range=(np.float32(0.5), np.float32(10.0)))
    assert pred["depth"].shape == (1, 128 // 4, 160 // 4)
    ims, intr = O.resize_inputs(s["images"], s["intrinsics"], 128, 160)
    shift, scl = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    norm = [np.transpose((np.transpose(im / 255.0, [1, 2, 0]) - shift) / scl, [2, 0, 1]).astype(np.float32)[None] for im in ims]
    ref = PL.mvsnet_forward(norm, [p[None] for p in s["poses"]], [k[None] for k in intr], 0, (0.5, 10.0), sd, D)
    np.testing.assert_allclose(pred["depth"], ref["depth"][0], rtol=1e-3)
