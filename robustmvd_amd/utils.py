"""Host-side helpers of the model protocol (numpy <-> torch conversion, batching, view selection).
Behaviour follows rmvd/utils/utils.py:126-347 of the reference (to_torch, to_numpy, numpy_collate,
select_by_index, exclude_index, get_torch_model_device); written for modern torch (no torch._six).
"""
import collections.abc as cabc

import numpy as np
import torch

_STR = (str, bytes)


def _is_np(x):
    return isinstance(x, (np.ndarray, np.generic)) and not isinstance(x, (np.str_, np.bytes_))


def _map(fn, data, leaf):
    """Applies fn to every leaf accepted by `leaf` inside nested dict / list / tuple containers.
    Plain tuples become lists (the reference's 'backwards compatibility' rule); namedtuples keep their type."""
    if data is None:
        return None
    if leaf(data):
        return fn(data)
    if isinstance(data, cabc.Mapping):
        return {k: _map(fn, v, leaf) for k, v in data.items()}
    if isinstance(data, tuple) and hasattr(data, "_fields"):
        return type(data)(*(_map(fn, v, leaf) for v in data))
    if isinstance(data, cabc.Sequence) and not isinstance(data, _STR):
        return [_map(fn, v, leaf) for v in data]
    return data


def to_torch(data, device=None):
    def conv(x):
        if isinstance(x, torch.Tensor):
            return x.to(device)
        if isinstance(x, np.ndarray) and x.dtype.kind in "SaUO":
            return x
        return torch.as_tensor(x, device=device)

    return _map(conv, data, lambda x: isinstance(x, torch.Tensor) or _is_np(x))


def to_numpy(data):
    return _map(lambda t: t.detach().cpu().numpy(), data, lambda x: isinstance(x, torch.Tensor))


def numpy_collate(batch):
    """Stacks a list of samples into one batched sample (leading axis), recursing into containers."""
    if batch is None:
        return None
    first = batch[0]
    if first is None:
        assert all(b is None for b in batch)
        return None
    if isinstance(first, torch.Tensor):
        return numpy_collate([b.detach().cpu().numpy() for b in batch])
    if isinstance(first, np.ndarray):
        return np.stack(batch, 0)
    if isinstance(first, (np.generic, float, int)) and not isinstance(first, (np.str_, np.bytes_)):
        return np.array(batch)
    if isinstance(first, _STR):
        return batch
    if isinstance(first, cabc.Mapping):
        return {k: numpy_collate([b[k] for b in batch]) for k in first}
    if isinstance(first, tuple) and hasattr(first, "_fields"):
        return type(first)(*(numpy_collate(list(s)) for s in zip(*batch)))
    if isinstance(first, cabc.Sequence):
        if any(len(b) != len(first) for b in batch):
            raise RuntimeError("each element in list of batch should be of equal size")
        return [numpy_collate(list(s)) for s in zip(*batch)]
    raise TypeError(f"numpy_collate: unsupported element type {type(first)}")


def get_torch_model_device(model):
    devices = {p.device for p in model.parameters()}
    if len(devices) != 1:
        raise RuntimeError("All model parameters need to be on the same device")
    return devices.pop()


def _stack(items):
    return np.stack(items, 0) if isinstance(items[0], np.ndarray) else torch.stack(items, 0)


def select_by_index(views, idx):
    """views: list over views of (batched) items; idx: int, or per-sample indices (N,)."""
    if isinstance(idx, (int, np.integer)):
        return views[int(idx)]
    return _stack([views[int(i)][b] for b, i in enumerate(idx)])


def exclude_index(views, idx):
    """All views except `idx` (int or per-sample indices); order of the remaining views is kept."""
    if isinstance(idx, (int, np.integer)):
        return [v for i, v in enumerate(views) if i != int(idx)]
    per_sample = [[v[b] for i, v in enumerate(views) if i != int(ex)] for b, ex in enumerate(idx)]
    if per_sample and all(len(r) > 0 for r in per_sample):
        return [_stack(list(col)) for col in zip(*per_sample)]
    return per_sample
