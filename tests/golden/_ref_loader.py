"""Load the hot-path modules of the reference (/root/reference) on CPU, file by file.

Used ONLY by tests/golden/make_golden.py in the build container to generate the
golden vectors.  Nothing here runs on the GPU box (the reference does not travel)
and nothing in the product imports it.

`import rmvd` fails on this image (pytoml / torch._six / kornia / torchvision /
skimage / cv2 are absent), so the handful of third-party names the hot path touches
are provided as tiny stand-in modules and each reference file is loaded with
importlib under its real dotted name.
"""
import importlib.util
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("RMVD_REFERENCE", "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name):
    m = _mod(name)
    m.__path__ = []
    return m


def _load(name, relpath):
    path = os.path.join(REF_ROOT, relpath)
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    m.__package__ = name.rpartition(".")[0]
    sys.modules[name] = m
    spec.loader.exec_module(m)
    parent, _, leaf = name.rpartition(".")
    if parent in sys.modules:
        setattr(sys.modules[parent], leaf, m)
    return m


_loaded = None


def load_reference():
    """Returns a namespace with the reference's hot-path modules."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference not found at {REF_ROOT}")

    # --- third-party stand-ins -------------------------------------------------
    _mod("torch._six", string_classes=(str, bytes))
    _mod("pytoml", load=lambda f: {})

    def create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=torch.float32):
        assert not normalized_coordinates
        xs = torch.arange(width, dtype=dtype, device=device)
        ys = torch.arange(height, dtype=dtype, device=device)
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        return torch.stack((xx, yy), -1).unsqueeze(0)  # 1,H,W,2 ; [...,0]=x

    kutils = _mod("kornia.utils", create_meshgrid=create_meshgrid)
    _pkg("kornia").utils = kutils

    class _Noop:
        def __init__(self, *a, **k):
            pass

    tvt = _mod("torchvision.transforms", Compose=_Noop, ToTensor=_Noop, Normalize=_Noop)
    _pkg("torchvision").transforms = tvt

    # --- package skeleton ------------------------------------------------------
    _pkg("rmvd")
    _pkg("rmvd.models")
    _pkg("rmvd.models.blocks")
    _pkg("rmvd.data")
    _mod(
        "rmvd.data.transforms",
        ResizeInputs=_Noop,
        UpscaleInputsToNextMultipleOf=_Noop,
        NormalizeImagesToMinMax=_Noop,
        NormalizeImagesByShiftAndScale=_Noop,
    )
    uu = _load("rmvd.utils.utils", "rmvd/utils/utils.py")
    up = _pkg("rmvd.utils")
    for k, v in uu.__dict__.items():
        if not k.startswith("_"):
            setattr(up, k, v)
    up.utils = uu
    sys.modules["rmvd.utils.utils"] = uu
    up.logging = _load("rmvd.utils.logging", "rmvd/utils/logging.py")

    ns = types.SimpleNamespace()
    b = "rmvd/models/blocks/"
    ns.blocks_utils = _load("rmvd.models.blocks.utils", b + "utils.py")
    ns.planesweep_corr = _load("rmvd.models.blocks.planesweep_corr", b + "planesweep_corr.py")
    ns.learned_fusion = _load("rmvd.models.blocks.learned_fusion", b + "learned_fusion.py")
    for n in ("dispnet_encoder", "dispnet_context_encoder", "dispnet_costvolume_encoder", "dispnet_decoder"):
        setattr(ns, n, _load("rmvd.models.blocks." + n, b + n + ".py"))
    ns.mvsnet_components = _load("rmvd.models.blocks.mvsnet_components", b + "mvsnet_components.py")
    ns.registry = _load("rmvd.models.registry", "rmvd/models/registry.py")
    ns.helpers = _load("rmvd.models.helpers", "rmvd/models/helpers.py")
    ns.robust_mvd = _load("rmvd.models.robust_mvd", "rmvd/models/robust_mvd.py")
    ns.mvsnet = _load("rmvd.models.mvsnet", "rmvd/models/mvsnet.py")
    ns.utils = uu
    _loaded = ns
    return ns
