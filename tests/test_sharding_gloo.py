"""N>1 path on CPU: two gloo ranks shard a frame list round-robin, each runs its frames independently
(here through the CPU oracle, since there is no GPU in this container), and the union of the per-rank
results equals the single-process result — no collective touches the data path; the process group only
carries the barrier / max-time reduction that bench.py uses."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _frame_result(idx):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from oracle import mvd_oracle as O
    rng = np.random.default_rng(idx)
    cost = rng.standard_normal((1, 8, 6, 7)).astype(np.float32) * 3
    depth, conf, _ = O.softmax_regress(cost, np.linspace(0.5, 10, 8, dtype=np.float32)[None])
    return depth


def _worker(rank, world, port, nframes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    from robustmvd_amd.sharding import frames_for_rank, timed_region
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = frames_for_rank(nframes, rank, world)
    out = {}

    def work():
        for i in mine:
            out[i] = _frame_result(i)

    dt = timed_region(work, sync=lambda: None, dist=dist, device=torch.device("cpu"))
    # the max-over-ranks time is identical on every rank
    t = torch.tensor([dt], dtype=torch.float64)
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    assert all(abs(float(g) - dt) < 1e-12 for g in gathered)
    q.put((rank, mine, {k: v.tolist() for k, v in out.items()}, dt))
    dist.destroy_process_group()


def test_two_rank_frame_sharding():
    from robustmvd_amd.sharding import frames_for_rank
    nframes, world = 7, 2
    assert frames_for_rank(7, 0, 2) == [0, 2, 4, 6] and frames_for_rank(7, 1, 2) == [1, 3, 5]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nframes, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = {}
    for rank, mine, out, dt in results:
        assert dt > 0
        for k, v in out.items():
            assert int(k) not in seen, "a frame was processed by two ranks"
            seen[int(k)] = np.asarray(v, dtype=np.float32)
    assert sorted(seen) == list(range(nframes))
    for i in range(nframes):
        np.testing.assert_array_equal(seen[i], _frame_result(i))
