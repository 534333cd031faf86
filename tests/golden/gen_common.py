"""Deterministic input/weight generators shared by make_golden.py (which feeds them to the
imported reference) and by the tests (which feed the SAME values to the oracle and to the HIP
path).  Only numpy's PCG64 `default_rng` is used, so values are reproducible across machines;
nothing here touches the reference.
"""
import numpy as np


def rng_array(seed, shape, scale=1.0, dtype=np.float32):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(dtype)


def fill_state_dict(shapes, seed):
    """shapes: {key: shape}.  Returns {key: float32 ndarray}, keys visited in sorted order.

    conv / deconv weights ~ N(0, 2/(1.04*fan_in)); biases ~ N(0, 0.1); BN weight and
    running_var ~ U(0.5, 1.5); BN bias and running_mean ~ N(0, 0.1).
    """
    rng = np.random.default_rng(seed)
    out = {}
    for key in sorted(shapes):
        shape = tuple(shapes[key])
        if key.endswith("num_batches_tracked"):
            continue
        if key.endswith("running_var") or (key.endswith("weight") and len(shape) == 1):
            v = rng.uniform(0.5, 1.5, shape)
        elif key.endswith("running_mean") or key.endswith("bias"):
            v = rng.standard_normal(shape) * 0.1
        else:
            fan_in = int(np.prod(shape[1:]))
            v = rng.standard_normal(shape) * np.sqrt(2.0 / (1.04 * fan_in))
        out[key] = v.astype(np.float32)
    return out


def robustmvd_weights(shapes, seed):
    """fill_state_dict + a positive offset on the inverse-depth channel of every prediction head:
    with zero-mean random weights relu(ch0) is 0 almost everywhere, which would leave the final
    depth map a constant and the end-to-end comparison blind."""
    sd = fill_state_dict(shapes, seed)
    for k in sd:
        if k.startswith("decoder.pred_") and k.endswith(".bias"):
            sd[k][0] += 1.0
    return sd


def rot_xyz(ax, ay, az):
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def synthetic_pose(rng, rot_sigma=0.05, trans_sigma=0.15):
    """source_to_key 4x4 (p_src = T p_key): SURVEY.md 8(d) synthetic poses."""
    T = np.eye(4)
    T[:3, :3] = rot_xyz(*(rng.standard_normal(3) * rot_sigma))
    T[:3, 3] = rng.standard_normal(3) * trans_sigma
    return T.astype(np.float32)


def synthetic_intrinsics(H, W):
    return np.array([[0.9 * W, 0, W / 2], [0, 0.9 * W, H / 2], [0, 0, 1]], dtype=np.float32)


def synthetic_sample(frame_idx, H, W, V):
    """One synthetic frame as the bench and the scaling tests use it."""
    rng = np.random.default_rng(frame_idx)
    images = [rng.uniform(0, 255, (3, H, W)).astype(np.float32) for _ in range(V + 1)]
    K = synthetic_intrinsics(H, W)
    poses = [np.eye(4, dtype=np.float32)] + [synthetic_pose(rng) for _ in range(V)]
    return {"images": images, "intrinsics": [K.copy() for _ in range(V + 1)], "poses": poses, "keyview_idx": 0}
