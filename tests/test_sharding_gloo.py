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


def _bench_worker(rank, world, port, q, fail_rank):
    """runs bench.py's own main() on a gloo rank with the model stubbed (MVD_BENCH_STUB=1)"""
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "LOCAL_RANK": str(rank),
                       "WORLD_SIZE": str(world), "MVD_BENCH_STUB": "1", "MVD_BENCH_SETTLE": "1"})
    if fail_rank is not None:
        os.environ["MVD_BENCH_FAIL_H2D_RANK"] = str(fail_rank)
    sys.path.insert(0, ROOT)
    import io
    import contextlib
    import bench
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1"])
    q.put((rank, out, buf.getvalue()))


def _run_bench(world, fail_rank=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q, fail_rank)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=180) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


def test_bench_main_control_flow_on_two_gloo_ranks():
    """bench.py's main(): rank environment, round-robin frames, barrier-bracketed timed region with the max-over-ranks time,
    ONE JSON line from rank 0 with the whole-job value, and the h2d_inclusive block at world 2 — the model call stubbed."""
    import json
    (r0, out0, text0), (r1, out1, text1) = _run_bench(2)
    assert text1.strip() == ""                       # only rank 0 prints
    lines = [l for l in text0.splitlines() if l.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["frames_of_rank0"] == [0, 2] and out1["frames_of_rank0"] == [1, 3]   # frame = rank + world * i
    # both ranks hold the same max-over-ranks time: value = world * steps / time
    assert abs(out0["ms_per_step"] - out1["ms_per_step"]) < 1e-9
    assert abs(line["value"] - 2 * 3 / (line["ms_per_step"] * 3e-3)) < 1e-6 * line["value"]
    assert "value" in line["h2d_inclusive"] and "stub" in line


def test_bench_h2d_block_survives_a_lone_failing_rank():
    """one rank fails while setting up h2d_inclusive: both skip the block (all-reduced ok flag), nobody hangs, the headline stays"""
    (r0, out0, _), (r1, out1, _) = _run_bench(2, fail_rank=1)
    assert "skipped" in out0["h2d_inclusive"] and "skipped" in out1["h2d_inclusive"]
    assert out0["value"] > 0
