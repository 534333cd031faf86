"""Model registry + factory: the drop-in boundary of the reference
(rmvd/models/registry.py:11-53, rmvd/models/factory.py:8-65, rmvd/models/helpers.py:21-172).

    model = create_model("robust_mvd", weights="robustmvd_600k.pt", num_gpus=1)
    pred, aux = model.run(images=[...], keyview_idx=0, poses=[...], intrinsics=[...])

Differences from the reference, all on the deployment side:
  * num_gpus selects ONE device per process (the engine scales as one process per GPU with no
    collectives); num_gpus > 1 is rejected instead of wrapping in nn.DataParallel;
  * pretrained weights are URL-only in the reference; offline they must be given as a file via `weights`.
"""
import collections.abc as cabc

import numpy as np
import torch

from .utils import numpy_collate

_entrypoints = {}
_trainable = {}


def register_model(arg=None, trainable=True):
    def deco(fn):
        _entrypoints[fn.__name__] = fn
        if trainable:
            _trainable[fn.__name__] = fn
        return fn

    return deco(arg) if callable(arg) else deco


def list_models(trainable_only=False):
    return sorted(_trainable if trainable_only else _entrypoints)


def has_model(name, trainable_only=False):
    return name in (_trainable if trainable_only else _entrypoints)


def get_model(name):
    if name not in _entrypoints:
        raise AssertionError(f'The requested model "{name}" does not exist. Available models are: {" ".join(list_models())}')
    return _entrypoints[name]


def add_batch_dim(images, keyview_idx, poses=None, intrinsics=None, depth_range=None):
    return numpy_collate([(images, keyview_idx, poses, intrinsics, depth_range)])


def remove_batch_dim(batch):
    if batch is None:
        return None
    if isinstance(batch, np.ndarray):
        return batch[0]
    if isinstance(batch, cabc.Mapping):
        return {k: remove_batch_dim(v) for k, v in batch.items()}
    if isinstance(batch, tuple) and hasattr(batch, "_fields"):
        return type(batch)(*(remove_batch_dim(v) for v in batch))
    if isinstance(batch, tuple):
        return tuple(remove_batch_dim(v) for v in batch)
    if isinstance(batch, cabc.Sequence) and not isinstance(batch, (str, bytes)):
        return [remove_batch_dim(v) for v in batch]
    raise TypeError(f"remove_batch_dim: unsupported type {type(batch)}")


def add_run_function(model):
    """model.run(images, keyview_idx, poses, intrinsics, depth_range): numpy in, numpy out, batched or not
    (helpers.py:65-89)."""

    @torch.no_grad()
    def run(images, keyview_idx, poses=None, intrinsics=None, depth_range=None, **_):
        unbatched = images[0].ndim == 3
        if unbatched:
            images, keyview_idx, poses, intrinsics, depth_range = add_batch_dim(images, keyview_idx, poses, intrinsics,
                                                                                depth_range)
        sample = model.input_adapter(images=images, keyview_idx=keyview_idx, poses=poses, intrinsics=intrinsics,
                                     depth_range=depth_range)
        pred, aux = model.output_adapter(model(**sample))
        if unbatched:
            pred, aux = remove_batch_dim((pred, aux))
        return pred, aux

    model.run = run
    return model


def _place(model, train, num_gpus):
    model.train() if train else model.eval()
    if num_gpus > 1:
        raise ValueError("num_gpus > 1: run one process per GPU (python -m torch.distributed.run ...); this engine "
                         "shards frames across processes instead of wrapping the model in nn.DataParallel")
    if num_gpus == 1:
        model = model.cuda()
    return model


def build_model_with_cfg(model_cls, cfg=None, weights=None, train=False, num_gpus=1, **kwargs):
    """helpers.py:104-172: construct, load a checkpoint {'model_state_dict': ...} (strict, 'module.' stripped),
    set mode, place on the device."""
    kw = dict(cfg or {})
    kw.update(kwargs)
    model = model_cls(**kw)
    if weights is not None:
        if str(weights).startswith("http"):
            raise RuntimeError(f"weights URL {weights}: no network access; download the file and pass its path")
        ckpt = torch.load(weights, map_location="cpu", weights_only=True)
        state = {k.replace("module.", ""): v for k, v in ckpt["model_state_dict"].items()}
        model.load_state_dict(state, strict=True)
    return _place(model, train, num_gpus)


def create_model(name, pretrained=True, weights=None, train=False, num_gpus=1, **kwargs):
    model = get_model(name)(pretrained=pretrained, weights=weights, train=train, num_gpus=num_gpus, **kwargs)
    add_run_function(model)
    model.name = name
    return model


def prepare_custom_model(model, train=False, num_gpus=1):
    assert not isinstance(model, torch.nn.DataParallel)
    return add_run_function(_place(model, train, num_gpus))
