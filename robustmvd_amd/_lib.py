"""ctypes binding of libmvd_hip.so (include/mvd.h).  PyTorch tensors are only used for device
memory and the current stream; the library sees raw device pointers.

There is NO CPU fallback: if the library is missing or cannot be loaded the import of any op raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmvd_hip.so")
# `make -C robustmvd_amd/csrc exp`: the same engine + experimental kernel variants selected by MVD_K3_CFG / MVD_K4_*.
# Never loaded by the product path; tools/ and the variants test route calls through it with use_experiments_library().
EXP_LIB_PATH = os.path.join(_HERE, "lib_exp", "libmvd_hip_exp.so")

MVD_MAX_VIEWS = 32
LAYOUT_NCDHW = 0
LAYOUT_NDHWC = 1
GRID_EXACT = 0x100
FEAT_NHWC_BORDER = 0x200
CONV3D_STRIDE1 = 0
CONV2D, DECONV2D, CONV2D_IMAGE = 0, 1, 2  # mvd_conv2d_split_f32 modes
CONV3D_STRIDE2 = 1
DECONV3D_STRIDE2 = 2
LAYOUT_NCHW = 0
LAYOUT_NHWC = 1
LAYOUT_NHWC_BORDER = 2
INVDEPTH_SHARED, INVDEPTH_BATCHED, INVDEPTH_PER_PIXEL = 0, 1, 2
REDUCE_VARIANCE = 0
REDUCE_VARIANCE_KEYSQ = 1
REDUCE_GROUPCORR = 2

_c_float_p = ctypes.c_void_p
_pp = ctypes.POINTER(ctypes.c_void_p)
_i = ctypes.c_int
_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/mvd.h one to one
SIGNATURES = {
    "mvd_version": (_i, []),
    "mvd_last_error": (ctypes.c_char_p, []),
    "mvd_sweep_corr_workspace_bytes": (_sz, [_i] * 7),
    "mvd_sweep_corr_f32": (_i, [_c_float_p, _pp, _c_float_p, _pp, _pp, _c_float_p, _i] + [_i] * 8
                           + [_pp, _pp, ctypes.c_void_p, _sz, ctypes.c_void_p]),
    "mvd_sweep_corr_ex_f32": (_i, [_c_float_p, _pp, _c_float_p, _pp, _pp, _c_float_p, _i, ctypes.c_float] + [_i] * 8
                              + [_pp, _pp, ctypes.c_void_p, _sz, ctypes.c_void_p]),
    "mvd_sweep_corr_nhwc_f32": (_i, [_c_float_p, _pp, _c_float_p, _pp, _pp, _c_float_p, _i, ctypes.c_float] + [_i] * 8
                                + [_pp, _pp, _i, _c_float_p, ctypes.c_void_p]),
    "mvd_fuse_views_nhwc_f32": (_i, [_pp, _pp, _pp] + [_i] * 6 + [_c_float_p, _c_float_p, _i, _c_float_p, ctypes.c_void_p]),
    "mvd_upsample2x_nhwc_f32": (_i, [_c_float_p, _c_float_p, _c_float_p] + [_i] * 5 + [ctypes.c_void_p]),
    "mvd_sweep_warp_f32": (_i, [_pp, _c_float_p, _pp, _pp, _c_float_p] + [_i] * 10 + [_pp, _pp, ctypes.c_void_p]),
    "mvd_fuse_views_f32": (_i, [_pp, _pp, _pp, _i, _i, _i, _i, _i, _c_float_p, _c_float_p, ctypes.c_void_p]),
    "mvd_warp_variance_workspace_bytes": (_sz, [_i] * 5),
    "mvd_warp_variance_f32": (_i, [_c_float_p, _pp, _pp, _c_float_p, _c_float_p] + [_i] * 6
                              + [_c_float_p, _i, ctypes.c_void_p, _sz, ctypes.c_void_p]),
    "mvd_warp_variance_absmax_f32": (_i, [_c_float_p, _pp, _pp, _c_float_p, _c_float_p] + [_i] * 6
                                     + [_c_float_p, _c_float_p, _i, ctypes.c_void_p, _sz, ctypes.c_void_p]),
    "mvd_absmax_f32": (_i, [_c_float_p, ctypes.c_longlong, _c_float_p, ctypes.c_void_p]),
    "mvd_warp_variance_f16_workspace_bytes": (_sz, [_i]),
    "mvd_warp_variance_f16": (_i, [ctypes.c_void_p, _pp, _pp, _c_float_p, _c_float_p] + [_i] * 5
                              + [ctypes.c_void_p, ctypes.c_void_p, _sz, ctypes.c_void_p]),
    "mvd_convert_f32_to_f16": (_i, [_c_float_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]),
    "mvd_convert_f16_to_f32": (_i, [ctypes.c_void_p, _c_float_p, ctypes.c_longlong, ctypes.c_void_p]),
    "mvd_homo_warp_f32": (_i, [_c_float_p] * 4 + [_i] * 5 + [_c_float_p, ctypes.c_void_p, _sz, ctypes.c_void_p]),
    "mvd_conv3d_packed_weight_floats": (_sz, [_i, _i]),
    "mvd_pack_conv3d_weights_f32": (_i, [_c_float_p, _i, _i, _i, _c_float_p, ctypes.c_void_p]),
    "mvd_conv3d_bn_relu_f32": (_i, [_c_float_p] * 6 + [_i] * 8 + [ctypes.c_void_p]),
    "mvd_conv3d_bn_relu_absmax_f32": (_i, [_c_float_p] * 7 + [_i] * 8 + [ctypes.c_void_p]),
    "mvd_conv3d_f16_packed_weight_bytes": (_sz, [_i, _i]),
    "mvd_pack_conv3d_weights_f16": (_i, [_c_float_p, _i, _i, ctypes.c_void_p, ctypes.c_void_p]),
    "mvd_conv3d_bn_relu_f16in": (_i, [ctypes.c_void_p, ctypes.c_void_p, _c_float_p, _c_float_p, _c_float_p] + [_i] * 7
                                 + [ctypes.c_void_p]),
    "mvd_conv3d_split_packed_weight_bytes": (_sz, [_i, _i]),
    "mvd_conv3d_bn_relu_absmax_f32_split": (_i, [_c_float_p, _c_float_p, ctypes.c_void_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p]
                                            + [_i] * 7 + [ctypes.c_void_p]),
    "mvd_conv3d_igemm_packed_weight_bytes": (_sz, [_i] * 3),
    "mvd_pack_conv3d_weights_igemm": (_i, [_c_float_p, _i, _i, _i, ctypes.c_void_p, ctypes.c_void_p]),
    "mvd_conv3d_igemm_workspace_bytes": (_sz, [_i] * 7),
    "mvd_conv3d_bn_relu_igemm_f32": (_i, [_c_float_p, _c_float_p, ctypes.c_void_p] + [_c_float_p] * 5 + [_i] * 8
                                     + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "mvd_conv2d_split_packed_weight_bytes": (_sz, [_i] * 6),
    "mvd_pack_conv2d_weights_split": (_i, [_c_float_p] + [_i] * 7 + [ctypes.c_void_p, ctypes.c_void_p]),
    "mvd_conv2d_split_workspace_bytes": (_sz, [_i] * 9),
    "mvd_conv2d_split_f32": (_i, [_c_float_p, _c_float_p, ctypes.c_void_p, _c_float_p, _c_float_p, _c_float_p] + [_i] * 7
                             + [ctypes.c_longlong] * 3 + [_i] * 5 + [ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "mvd_pack_conv3d_weights_split": (_i, [_c_float_p, _i, _i, ctypes.c_void_p, ctypes.c_void_p]),
    "mvd_conv3d_bn_relu_f32_split": (_i, [_c_float_p, _c_float_p, ctypes.c_void_p, _c_float_p, _c_float_p, _c_float_p] + [_i] * 7
                                     + [ctypes.c_void_p]),
    "mvd_conv2d_packed_weight_floats": (_sz, [_i, _i, _i]),
    "mvd_pack_conv2d_weights_f32": (_i, [_c_float_p, _i, _i, _i, _c_float_p, ctypes.c_void_p]),
    "mvd_conv2d_bn_relu_f32": (_i, [_c_float_p, _i, _c_float_p, _c_float_p, _c_float_p, _c_float_p, _i] + [_i] * 8
                               + [ctypes.c_void_p]),
    "mvd_conv2d_head_tile_count": (_sz, [_i] * 3),
    "mvd_conv2d_head_f32": (_i, [_c_float_p] * 9 + [_i] * 3 + [ctypes.c_void_p]),
    "mvd_conv2d_bn_relu_absmax_f32": (_i, [_c_float_p, _i, _c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p, _i] + [_i] * 8
                                      + [ctypes.c_void_p]),
    "mvd_softmax_regress_f32": (_i, [_c_float_p, _c_float_p, _i, _i, _i, _i, _c_float_p, _c_float_p, ctypes.c_void_p]),
    "mvd_bias_leaky_relu_f32": (_i, [_c_float_p, _c_float_p, _i, _i, ctypes.c_longlong, ctypes.c_float, ctypes.c_void_p]),
    "mvd_dispnet_head_f32": (_i, [_c_float_p, _c_float_p, _c_float_p, _i, ctypes.c_longlong, ctypes.c_void_p]),
    "mvd_arm_kernel_timing": (_i, [ctypes.c_void_p, ctypes.c_void_p]),
    "mvd_stream_fill_f32": (_i, [_c_float_p, ctypes.c_longlong, ctypes.c_float, ctypes.c_void_p]),
    "mvd_warp_variance_backward_workspace_bytes": (_sz, [_i]),
    "mvd_warp_variance_backward_f32": (_i, [_c_float_p, _pp, _pp, _c_float_p, _c_float_p, _c_float_p] + [_i] * 6
                                       + [_c_float_p, _pp, ctypes.c_void_p, _sz, ctypes.c_void_p]),
    "mvd_sweep_corr_backward_f32": (_i, [_c_float_p, _pp, _c_float_p, _pp, _pp, _c_float_p, _i, ctypes.c_float, _pp] + [_i] * 8
                                    + [_c_float_p, _pp, ctypes.c_void_p]),
    "mvd_fuse_views_backward_f32": (_i, [_pp, _pp, _pp, _c_float_p] + [_i] * 5 + [_pp, _pp, ctypes.c_void_p]),
    "mvd_sweep_reduce_workspace_bytes": (_sz, [_i] * 5),
    "mvd_sweep_reduce_f32": (_i, [_c_float_p, _pp, _pp, _c_float_p, _i] + [ctypes.c_float] * 4 + [_i] * 8
                             + [_pp, ctypes.c_void_p, _sz, ctypes.c_void_p]),
    "mvd_resize_order1_f32": (_i, [_c_float_p, _c_float_p, ctypes.c_longlong, _i, _i, _i, _i, ctypes.c_void_p]),
    "mvd_nchw_to_nhwc_f32": (_i, [_c_float_p, _c_float_p, _i, _i, ctypes.c_longlong, ctypes.c_void_p]),
    "mvd_nhwc_to_nchw_f32": (_i, [_c_float_p, _c_float_p, _i, _i, ctypes.c_longlong, ctypes.c_void_p]),
}

_lib = None
_override = None  # set by use_experiments_library()


def _bind(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """Loads libmvd_hip.so (once).  Raises RuntimeError with build instructions if it is absent."""
    global _lib
    if _override is not None:
        return _override
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP engine is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C robustmvd_amd/csrc`. There is no CPU fallback for this path.")
    _lib = _bind(LIB_PATH)
    return _lib


class use_experiments_library:
    """Context manager for tools/ and tests: inside it every op goes through libmvd_hip_exp.so (path overridable),
    whose kernel variants MVD_K3_CFG / MVD_K4_* select.  The product library is untouched and stays loaded."""

    def __init__(self, path=None):
        self.path = path or EXP_LIB_PATH

    def __enter__(self):
        global _override
        if not os.path.exists(self.path):
            raise RuntimeError(f"{self.path} not found: run `make -C robustmvd_amd/csrc exp`")
        self._prev = _override
        _override = _bind(self.path)
        return _override

    def __exit__(self, *exc):
        global _override
        _override = self._prev
        return False


def check(rc, what):
    if rc != 0:
        msg = load().mvd_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (status {rc}): {msg}")


def stream_of(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return ctypes.cast(arr, _pp), arr  # keep `arr` alive in the caller


def as_f16(t, name, shape=None, device=None):
    """Like as_f32 for the fp16 entry points: no silent conversion, the caller decides where the rounding happens."""
    if not isinstance(t, torch.Tensor):
        raise ValueError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise ValueError(f"{name}: tensor is on {t.device}; the HIP engine needs a cuda (ROCm) device tensor")
    if device is not None and t.device != device:
        raise ValueError(f"{name}: on {t.device}, expected {device}")
    if t.dtype != torch.float16:
        raise ValueError(f"{name}: dtype {t.dtype}, expected torch.float16")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t.contiguous()


def as_f32(t, name, shape=None, device=None):
    """Validates a tensor argument at the boundary (reference: asserts, planesweep_corr.py:444,473-483)."""
    if not isinstance(t, torch.Tensor):
        raise ValueError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise ValueError(f"{name}: tensor is on {t.device}; the HIP engine needs a cuda (ROCm) device tensor")
    if device is not None and t.device != device:
        raise ValueError(f"{name}: on {t.device}, expected {device}")
    if t.dtype != torch.float32:
        t = t.float()
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t.contiguous()
