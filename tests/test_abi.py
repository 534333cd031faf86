"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/mvd.h declares (no compute call is made: there is no GPU here), and the Python binding's
signature table covers exactly those symbols."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mvd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mvd_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    syms = declared_symbols()
    for need in ("mvd_sweep_corr_f32", "mvd_fuse_views_f32", "mvd_warp_variance_f32", "mvd_conv3d_bn_relu_f32",
                 "mvd_softmax_regress_f32", "mvd_conv2d_bn_relu_f32", "mvd_pack_conv2d_weights_f32", "mvd_bias_leaky_relu_f32",
                 "mvd_version", "mvd_last_error"):
        assert need in syms


def test_library_exports_every_declared_symbol():
    from robustmvd_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/mvd.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    lib.mvd_version.restype = ctypes.c_int
    assert lib.mvd_version() == 100


def test_missing_library_fails_loudly(monkeypatch):
    from robustmvd_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmvd_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_argument_validation_without_gpu():
    import torch
    from robustmvd_amd import ops
    with pytest.raises(ValueError, match="cuda"):
        ops.softmax_regress(torch.zeros(1, 4, 2, 2), torch.zeros(1, 4))
    with pytest.raises(ValueError):
        ops.sweep_corr("nope", [], None, [], [], None)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "robustmvd_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_product_library_never_reads_the_environment():
    """include/mvd.h: "no global mutable state".  The shipped library must not import getenv and must not carry the
    experiment selectors' names; those exist only in the separate experiments build (`make exp`)."""
    import subprocess
    from robustmvd_amd import _lib
    und = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"MVD_K3_CFG", b"MVD_K4_NOMARCH", b"MVD_K4_NOKSPLIT", b"MVD_K4_DECONV_CLASSES", b"MVD_K4_MARCH_MIN"):
        assert name not in blob, f"{name.decode()} is compiled into the product library"
    for f in os.listdir(os.path.join(ROOT, "robustmvd_amd", "csrc")):
        if f.endswith((".hip", ".h")) and f != "mvd_api.hip":
            assert "getenv" not in open(os.path.join(ROOT, "robustmvd_amd", "csrc", f)).read(), f
