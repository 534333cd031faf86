#!/usr/bin/env python3
"""Headline benchmark: depth-maps/sec through the MVSNet-style plane-sweep path (warp+variance ->
3-D regulariser -> soft argmin, with its 2-D feature net) at the ETH3D shape 768x1152, 4 source views,
256 depth planes, fp32 (BASELINE.json configs[2]), on N GPUs of one node as independent replicas.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one forward pass (one key view + V sources -> one depth map) on inputs already resident in HBM.
Frames are sharded over the ranks (each rank runs K steps on its own frames, weak scaling); there is no
data-path collective — the process group is only used for the barrier and the max-over-ranks time.
Rank 0 prints ONE JSON line with the contract fields plus `roofline` (warp+variance kernel, HIP events
around every launch of it inside the timed region), `cpu_baseline` (the CPU oracle port timed on this
host, N=1 only) and `path_a` (the robust_mvd model at the same shape, same protocol).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)
import gen_common as gc  # noqa: E402

CONFIGS = {  # BASELINE.json configs (index -> H, W, V, D)
    1: (448, 640, 2, 128),
    2: (768, 1152, 4, 256),
    3: (896, 1216, 4, 256),
    4: (704, 1280, 6, 512),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
# untimed set-up forwards before the W warm-up steps of the contract (code objects, MIOpen solver search for path_a,
# allocator and clocks to steady state); reported in the JSON line as `settle_forwards`
SETTLE_FORWARDS = int(os.environ.get("MVD_BENCH_SETTLE", "12"))


# configs[3] is named "fp16 features" in BASELINE.json: features rounded to fp16, fp16 variance volume, regulariser's first
# layer on fp16 MFMA (fp32 accumulation); every other config is fp32 end to end
HALF_FEATURES = {3}


def build_mvsnet(D, dev, seed=0, half_features=False, conv0_split=True, exact_grid=False):
    import robustmvd_amd as R
    model = R.MVSNet(num_sampling_steps=D, half_features=half_features, conv0_split=conv0_split, exact_grid=exact_grid).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = gc.fill_state_dict(shapes, seed)
    full = model.state_dict()
    for k, v in sd.items():
        full[k] = torch.from_numpy(v)
    model.load_state_dict(full)
    return R.add_run_function(model.to(dev)), sd


def build_robustmvd(dev, seed=0, half_dispnet=False, engine_dispnet=True):
    import robustmvd_amd as R
    model = R.RobustMVD(half_dispnet=half_dispnet, engine_dispnet=engine_dispnet).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = gc.robustmvd_weights(shapes, seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return R.add_run_function(model.to(dev)), sd


def adapted_sample(model, frame_idx, H, W, V, depth_range=None):
    s = gc.synthetic_sample(frame_idx, H, W, V)
    from robustmvd_amd.registry import add_batch_dim
    images, key, poses, intr, dr = add_batch_dim(s["images"], s["keyview_idx"], s["poses"], s["intrinsics"], depth_range)
    return model.input_adapter(images=images, keyview_idx=key, poses=poses, intrinsics=intr, depth_range=dr)


def device_sync(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)


class StubModel:
    """MVD_BENCH_STUB=1 (tests/test_sharding_gloo.py): stands in for the engine so that two gloo ranks on CPU can run THIS
    script's control flow — rank environment, frame sharding, barrier-bracketed timed region, max-over-ranks time, rank-0 JSON
    line — without a GPU.  One "forward" reduces the frame's images."""

    def input_adapter(self, images, keyview_idx, poses=None, intrinsics=None, depth_range=None):
        return {"images": [torch.as_tensor(np.asarray(im)) for im in images]}

    def __call__(self, images):
        return {"depth": sum(float(im.double().mean()) for im in images)}, {}


def timed_loop(model, samples, steps, warmup, world, dev, arm=None, cdev=None):
    """W untimed + K timed forwards bracketed by barrier + synchronize; returns seconds (max over ranks)."""
    import torch.distributed as dist
    from robustmvd_amd.sharding import timed_region

    def run(n, armed):
        with torch.no_grad():
            for i in range(n):
                if armed and arm is not None:
                    arm(i)
                model(**samples[i % len(samples)])

    # set-up, not measurement: the first calls load code objects, run MIOpen's solver search for the adjacent 2-D
    # convolutions and bring the allocator and the clocks to steady state; then the W warm-up steps of the contract
    run(SETTLE_FORWARDS, False)
    device_sync(dev)
    run(warmup, False)
    return timed_region(lambda: run(steps, True), sync=lambda: device_sync(dev),
                        dist=dist if world > 1 else None, device=cdev or dev)


def measured_copy_gbs(dev):
    """Attainable HBM bandwidth on this box: device-to-device copy of 1 GiB (read + write bytes / time)."""
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(dev)
    return 2 * 4 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def measured_stream_gbs(dev, lib):
    """Attainable HBM WRITE bandwidth on this box: the library's streaming-store pass (mvd_stream_fill_f32) over 2 GiB —
    what a kernel bound by the write of its output (K3: 98 % of its algorithmic bytes) can reach at best."""
    from robustmvd_amd import _lib as L
    n = 1 << 29
    a = torch.empty(n, dtype=torch.float32, device=dev)
    st = L.stream_of(a)
    for _ in range(3):
        L.check(lib.mvd_stream_fill_f32(L.ptr(a), n, 1.0, st), "mvd_stream_fill_f32")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.mvd_stream_fill_f32(L.ptr(a), n, 1.0, st)
    e1.record()
    torch.cuda.synchronize(dev)
    return 4.0 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def cpu_baseline(H, W, V, D, sd, frames=3, burn_in=1):
    """The CPU oracle port (C/OpenMP hot path + torch-CPU 2-D feature net) on a bounded sample of the workload, with the
    protocol of BASELINE.md section 3: `burn_in` discarded forward(s), then the MEDIAN of `frames` timed full frames
    (one forward each; about 15-20 s of CPU work in total on the GPU box's host)."""
    from oracle import c_oracle as CO
    from oracle import pipeline as PL
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(3, 1, 1)
    times, stages = [], {}
    t_all = time.perf_counter()
    for f in range(burn_in + frames):
        s = gc.synthetic_sample(f, H, W, V)
        images = [((im / 255.0 - mean) / std).astype(np.float32)[None] for im in s["images"]]
        timings = {}
        t0 = time.perf_counter()
        PL.mvsnet_forward(images, [p[None] for p in s["poses"]], [k[None] for k in s["intrinsics"]], 0, (0.5, 10.0), sd, D,
                          timings=timings)
        if f < burn_in:
            continue
        times.append(time.perf_counter() - t0)
        for k, v in timings.items():
            stages.setdefault(k, []).append(v)
    total = time.perf_counter() - t_all
    dt = float(np.median(times))
    st = {k: float(np.median(v)) for k, v in stages.items()}
    return {"value": 1.0 / dt, "unit": "depth-maps/sec", "cores": CO.num_threads(), "kind": "port",
            "sample": f"{burn_in} burn-in + {frames} timed full frames {H}x{W} V{V} D{D} (one forward each; median {dt:.2f} s per "
                      f"frame, {total:.1f} s of CPU work in total; oracle/pipeline.py: C/OpenMP warp+variance "
                      f"{st['warp_variance']:.2f} s, CostRegNet {st['cost_reg']:.2f} s, soft-argmin {st['regress']:.2f} s, "
                      f"torch-CPU FeatureNet {st['features']:.2f} s)",
            "times_s": [round(t, 3) for t in times], "host_cpus": os.cpu_count(), "torch_threads": torch.get_num_threads()}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, help="index into BASELINE.json configs (default 2 = headline)")
    ap.add_argument("--fp32", action="store_true", help="run configs[3] on the fp32 path instead of its named fp16-feature variant")
    ap.add_argument("--include-h2d", action="store_true", help="(kept for compatibility: h2d_inclusive is on by default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-path-a", action="store_true")
    args = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    stub = os.environ.get("MVD_BENCH_STUB", "0") == "1"  # CPU rehearsal of the control flow (tests/test_sharding_gloo.py)
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback for the engine)")
    if stub:
        dev = torch.device("cpu")
        ndev = 0
    else:
        ndev = torch.cuda.device_count()
        dev = torch.device("cuda", local_rank % ndev)
        torch.cuda.set_device(dev)
    cdev = dev  # device of the tensors that go through the process group
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not stub and ndev >= world:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:  # rehearsal with more ranks than GPUs (RCCL refuses two ranks on one device), or the CPU stub: control plane on gloo
            dist.init_process_group("gloo", rank=rank, world_size=world)
            cdev = torch.device("cpu")

    # MIOpen's exhaustive solver search for the adjacent 2-D convolutions measured no gain here (and costs ~20 s of
    # naive-kernel trials at start-up), so it stays off unless asked for
    torch.backends.cudnn.benchmark = os.environ.get("MVD_BENCH_MIOPEN_FIND", "0") == "1"
    H, W, V, D = CONFIGS[args.config]
    if stub:
        H, W = 32, 48
    h, w, C = H // 4, W // 4, 32
    half = args.config in HALF_FEATURES and not args.fp32
    if stub:
        model, sd = StubModel(), None
    else:
        model, sd = build_mvsnet(D, dev, half_features=half)
    # this rank's frames: frame index = rank + world * i (round-robin shard of the frame list)
    nframes = 2
    from robustmvd_amd.sharding import frames_for_rank
    my_frames = frames_for_rank(nframes * world, rank, world)
    if stub:
        samples = [model.input_adapter(**{k: v for k, v in gc.synthetic_sample(f, H, W, V).items()}) for f in my_frames]
    else:
        samples = [adapted_sample(model, f, H, W, V, (np.float32(0.5), np.float32(10.0))) for f in my_frames]

    # HIP events around every warp+variance launch inside the timed region
    lib, ev = None, []
    if not stub:
        from robustmvd_amd import _lib as L
        lib = L.load()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a, b in ev:  # force creation of the underlying hipEvent_t
            a.record(); b.record()
        torch.cuda.synchronize(dev)

    def arm(i):
        if stub or os.environ.get("MVD_BENCH_NO_ARM"):
            return
        lib.mvd_arm_kernel_timing(ev[i][0].cuda_event, ev[i][1].cuda_event)

    dt = timed_loop(model, samples, args.steps, args.warmup, world, dev, arm, cdev)
    k3_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else float("nan")
    # SURVEY.md 8(d), batch 1 per launch; the fp16-feature variant moves 2-byte features and a 2-byte volume
    k3_bytes = (2.0 if half else 4.0) * ((V + 1) * C * h * w + C * D * h * w)
    value = world * args.steps / dt

    out = {
        # BASELINE.json's metric string, with the sizes of the config actually run (identical for the default config 2)
        "metric": f"depth-maps/sec at {H}x{W}x{V}srcx{D}planes; warp+aggregate HBM GB/s vs roofline",
        "value": value, "unit": "depth-maps/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 features + volume, f32 arithmetic (fp16-MFMA first regulariser layer, f32 accumulate)" if half else "f32",
        "data": "synthetic", "settle_forwards": SETTLE_FORWARDS, "frames_of_rank0": my_frames,
        "config": {"workload": f"mvsnet (Path B) forward {H}x{W}, {V} source views, {D} planes, batch 1 per step "
                               f"(BASELINE.json configs[{args.config}])" + ("" if half else
                               "; the regulariser's two layers with 32 input channels (conv0, conv4): every fp32 operand as two range-scaled fp16 terms "
                               "on fp16 MFMA, fp32 accumulation (at least as close to a float64 convolution as the fp32 matrix instruction "
                               "over 1e-42..1e30: tests/test_hip_f16.py), the other nine layers and everything else on fp32 arithmetic"),
                   "parallelism": f"{world} independent replica(s), frames round-robin, no collectives"},
    }
    if stub:
        out["stub"] = "MVD_BENCH_STUB=1: control-flow rehearsal on CPU, the model is a stand-in; not a measurement"
    if rank == 0 and not stub:
        # `traffic` is NOT measured in this run: PMC counters need rocprofv3 passes of their own.  It is the constant
        # the last committed counter passes gave for this config (profiles/k3_traffic.json, with its source files)
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "k3_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("config") == args.config:
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_src = "rocprofv3 PMC-derived constant, not measured in this run: " + str(tj.get("source", tpath))
        copy_gbs = measured_copy_gbs(dev)
        stream_gbs = measured_stream_gbs(dev, lib)
        achieved = k3_bytes / (k3_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "warp_variance_f16 (K3)" if half else "warp_variance (K3)", "achieved": achieved,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": k3_bytes,
                           "avg_launch_ms": k3_ms, "launches_timed": args.steps, "measured_stream_peak_gbs": stream_gbs,
                           "frac_of_measured_stream": achieved / stream_gbs, "measured_copy_peak_gbs": copy_gbs,
                           "frac_of_measured_copy": achieved / copy_gbs}
    # extra, not the headline: the same K steps with two frames in flight on separate HIP streams of this process
    # (kernels of one frame fill the matrix-pipe bubbles and tails of the other's); every step still is one batch-1
    # forward and all K complete inside the bracketed region
    def guarded(name, fn):
        """The extra blocks must not cost the headline line: on a single rank a failure is reported in place."""
        try:
            fn()
        except Exception as e:  # noqa: BLE001
            if world > 1:
                raise
            out[name] = {"error": repr(e)[:300]}

    def pipelined_block():
        import torch.distributed as dist
        from robustmvd_amd.sharding import timed_region
        import robustmvd_amd as R
        nfl = 2
        pipe = R.FramePipeline(model, depth=nfl)

        def run_pipelined(n):
            for i in range(n):
                pipe.submit(**samples[i % len(samples)])

        run_pipelined(args.warmup + 2)
        torch.cuda.synchronize(dev)
        dtp = timed_region(lambda: run_pipelined(args.steps), sync=lambda: torch.cuda.synchronize(dev),
                           dist=dist if world > 1 else None, device=cdev)
        out["pipelined"] = {"frames_in_flight": nfl, "value": world * args.steps / dtp, "unit": "depth-maps/sec",
                            "ms_per_step": dtp / args.steps * 1e3,
                            "note": "same workload and step count, two HIP streams per process; not the headline value"}

    # Multi-rank runs (the driver's scaling curve) measure the headline and h2d_inclusive (which guards itself against a lone
    # failing rank): the other extra blocks contain barriers too, so a failure on ONE rank would hang the others and lose the
    # whole line.  MVD_BENCH_EXTRAS=1 forces all of them on, =0 turns every extra off at one rank too (clean kernel statistics
    # under rocprofv3).
    extras = ((world == 1 and os.environ.get("MVD_BENCH_EXTRAS", "") != "0") or os.environ.get("MVD_BENCH_EXTRAS", "0") == "1") and not stub
    if extras and os.environ.get("MVD_BENCH_PIPELINED", "1") == "1":
        guarded("pipelined", pipelined_block)

    def h2d_inclusive_block():
        """Extra, not the headline (`value` keeps the contract: inputs resident in HBM): the same K steps with each
        frame's raw 0..255 images starting in PINNED HOST memory — H2D on a copy stream, overlapped with the previous
        frame's forward, then the model's input_adapter (normalisation on the device) and the forward.  This is what
        SURVEY.md 8(e) expects to limit 1 -> 8 GPU scaling (5 x 10.6 MB per frame at the headline shape), so multi-rank runs
        report it too.  Its timed region contains collectives: every rank first does its set-up and warm-up under try/except,
        the ranks agree (all-reduced MIN of an ok flag) and only then enter the timed region together — a failure on one rank
        skips the block everywhere instead of hanging the others."""
        import torch.distributed as dist
        from robustmvd_amd.sharding import timed_region
        run_h2d, err = None, None
        try:
            if stub:
                host = [gc.synthetic_sample(f, H, W, V) for f in my_frames]

                def run_h2d(n):
                    for i in range(n):
                        model(**model.input_adapter(**host[i % len(host)]))
            else:
                from robustmvd_amd.registry import add_batch_dim
                import robustmvd_amd as R
                host = []
                for f in my_frames:
                    smp = gc.synthetic_sample(f, H, W, V)
                    im, key, po, intr, dr = add_batch_dim(smp["images"], smp["keyview_idx"], smp["poses"], smp["intrinsics"],
                                                          (np.float32(0.5), np.float32(10.0)))
                    host.append((R.PinnedUploader.pin(im), key, po, intr, dr))
                up = R.PinnedUploader(dev)

                def run_h2d(n):
                    nxt = up.stage(host[0][0])
                    with torch.no_grad():
                        for i in range(n):
                            _, key, po, intr, dr = host[i % len(host)]
                            cur, nxt = nxt, up.stage(host[(i + 1) % len(host)][0])  # next frame's upload runs under this forward
                            model(**model.input_adapter(images=cur.wait(), keyview_idx=key, poses=po, intrinsics=intr, depth_range=dr))

            if os.environ.get("MVD_BENCH_FAIL_H2D_RANK", "") == str(rank):  # test hook: a lone failing rank
                raise RuntimeError("injected set-up failure (MVD_BENCH_FAIL_H2D_RANK)")
            run_h2d(args.warmup + 2)
            device_sync(dev)
        except Exception as e:  # noqa: BLE001
            err = repr(e)[:300]
        ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=cdev)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            out["h2d_inclusive"] = {"skipped": "set-up failed on at least one rank", "error_on_rank0": err}
            return
        dth = timed_region(lambda: run_h2d(args.steps), sync=lambda: device_sync(dev),
                           dist=dist if world > 1 else None, device=cdev)
        mb = (V + 1) * 3 * H * W * 4 / 1e6
        out["h2d_inclusive"] = {"value": world * args.steps / dth, "unit": "depth-maps/sec", "ms_per_step": dth / args.steps * 1e3,
                                "host_to_device_mb_per_step": mb,
                                "note": "raw images in pinned host memory -> copy stream (overlapped with the previous "
                                        "frame) -> input_adapter on the device -> forward; not the headline value"}

    # on by default at every rank count (MVD_BENCH_H2D=0 turns it off); it handles a failing rank itself, so it is not `guarded`
    if os.environ.get("MVD_BENCH_H2D", "1") == "1" and os.environ.get("MVD_BENCH_EXTRAS", "") != "0":
        h2d_inclusive_block()

    def fp32_conv0_block():
        """Extra, not the headline: the same forward with every regulariser layer on the fp32 matrix instruction
        (MVSNet(conv0_split=False), the round-2 headline arithmetic), and the difference of the two depth maps."""
        ms, _ = build_mvsnet(D, dev, half_features=half, conv0_split=False)
        with torch.no_grad():
            d0 = model(**samples[0])[0]["depth"].clone()
            d1 = ms(**samples[0])[0]["depth"].clone()
        torch.cuda.synchronize(dev)
        rel = float(((d1 - d0).abs() / d0.abs()).max())
        dts = timed_loop(ms, samples, args.steps, args.warmup, world, dev, None, cdev)
        out["conv0_fp32_mfma"] = {"value": world * args.steps / dts, "unit": "depth-maps/sec", "ms_per_step": dts / args.steps * 1e3,
                                  "max_rel_depth_diff_vs_headline_model": rel,
                                  "note": "MVSNet(conv0_split=False): conv0, conv2 and conv4 on v_mfma_f32_16x16x4_f32 like the other eight layers; "
                                          "not the headline value"}
        del ms

    if extras and not half:
        guarded("conv0_fp32_mfma", fp32_conv0_block)

    def exact_grid_block():
        """Extra, not the headline: MVSNet(exact_grid=True) — K3's sampling positions by the reference's own rounding chain
        (IEEE divisions, normalise then un-normalise: blocks/utils.py:234-266) instead of the folded form; with it every pixel
        of the full-size forward is within rtol 1e-3 of the oracle (tests/test_hip_configs.py).  Reports the step and K3's time."""
        mx, _ = build_mvsnet(D, dev, half_features=half, exact_grid=True)
        evx = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a, b in evx:
            a.record(); b.record()
        torch.cuda.synchronize(dev)
        dtx = timed_loop(mx, samples, args.steps, args.warmup, world, dev,
                         lambda i: lib.mvd_arm_kernel_timing(evx[i][0].cuda_event, evx[i][1].cuda_event), cdev)
        k3x = float(np.mean([a.elapsed_time(b) for a, b in evx]))
        out["exact_grid"] = {"value": world * args.steps / dtx, "unit": "depth-maps/sec", "ms_per_step": dtx / args.steps * 1e3,
                             "warp_variance_ms": k3x, "warp_variance_ms_headline": k3_ms,
                             "note": "MVSNet(exact_grid=True): reference rounding chain in K3's locate phase; not the headline value"}
        del mx

    if extras and not half:
        guarded("exact_grid", exact_grid_block)

    def mfma_roofline_block():
        """Second roofline block: the LONGEST kernel of the step, the regulariser's first layer (K4 conv0: 3x3x3, 32 -> 8) in its
        split-operand form, against the dense fp16 matrix peak: 2 MFMAs (16x16x32) per tap and 16 voxels.  Timed standalone
        (torch events around 10 back-to-back launches on a volume of the step's shape) after the timed region; the per-launch
        time agrees with its row in the rocprofv3 kernel stats."""
        from robustmvd_amd import ops
        h4, w4 = H // 4, W // 4
        reg = model.cost_regularization
        pk = reg._prepare()
        _, _, _, sc, sh, _ = pk["conv0"]
        x = torch.rand(1, D, h4, w4, 32, device=dev)
        amax = ops.absmax(x)
        for _ in range(3):
            y = ops.conv3d_bn_relu_split(x, pk["conv0_split"], sc, sh, relu=True, x_absmax=amax)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            y = ops.conv3d_bn_relu_split(x, pk["conv0_split"], sc, sh, relu=True, x_absmax=amax)
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / n
        flops = 2.0 * 27 * 32 * 8 * D * h4 * w4                      # the layer's arithmetic (fp32-equivalent)
        issued = 2.0 * 27 * (D * h4 * w4 / 16.0) * 2 * 16 * 16 * 32  # what the matrix pipe executes: 2 MFMAs per (tap, 16 voxels)
        out["roofline_mfma"] = {"bound": "mfma", "kernel": "conv3d first layer, split operands (K4 conv0, 32->8, 3x3x3)",
                                "achieved": issued / ms / 1e9, "peak": 2500.0, "unit": "TFLOP/s", "frac": issued / ms / 1e9 / 2500.0,
                                "issued_flops_per_launch": issued, "algorithmic_flops_per_launch": flops,
                                "fp32_equivalent_tflops": flops / ms / 1e9, "avg_launch_ms": ms, "launches_timed": n,
                                "note": "dense fp16 MFMA peak (MI355X_MICROARCH.md, ~2.5 PFLOP/s); three of the four operand products "
                                        "(hi*hi, hi*lo, lo*hi) in two MFMAs, one of them half used; the kernel is bound by staging "
                                        "(split conversion + LDS), not by the matrix pipe (DESIGN.md K4); timed standalone after the "
                                        "timed region"}
        del x, y

    if extras and not half:
        guarded("roofline_mfma", mfma_roofline_block)
    del samples
    if not stub:
        torch.cuda.empty_cache()

    def path_a_block():
        # robust_mvd (Path A, the create_model("robust_mvd") drop-in) at the same image shape; S = 256 planes fixed
        ma, _ = build_robustmvd(dev)
        sa = [adapted_sample(ma, f, H, W, V) for f in my_frames]
        eva = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a, b in eva:
            a.record(); b.record()
        torch.cuda.synchronize(dev)
        dta = timed_loop(ma, sa, args.steps, args.warmup, world, dev,
                         lambda i: lib.mvd_arm_kernel_timing(eva[i][0].cuda_event, eva[i][1].cuda_event), cdev)
        k1_ms = float(np.mean([a.elapsed_time(b) for a, b in eva]))
        hs, ws_ = H // 8, W // 8
        k1_bytes = 4.0 * ((V + 1) * 256 * hs * ws_ + 2 * V * 256 * hs * ws_)
        # K1 is not an HBM kernel: its roofline is vector-ALU issue / the texture addresser.  The busy
        # fractions are rocprofv3 PMC constants of the last committed passes (profiles/k1_pmc.json), not measured here.
        k1_pmc = None
        ppath = os.path.join(ROOT, "profiles", "k1_pmc.json")
        if os.path.exists(ppath) and args.config == 2:
            pj = json.load(open(ppath))
            k1_pmc = {"valu_busy_frac": pj.get("valu_busy_frac"), "ta_busy_frac": pj.get("ta_busy_frac"),
                      "source": "rocprofv3 PMC-derived constants, not measured in this run: " + str(pj.get("source"))}
        k1_flops = V * 256 * hs * ws_ * 256 * 10  # direct form: V*S*h*w*C*(2+8), SURVEY.md 8(d)
        out["path_a"] = {"model": "robust_mvd", "value": world * args.steps / dta, "unit": "depth-maps/sec",
                         "ms_per_step": dta / args.steps * 1e3, "sweep_corr_ms": k1_ms,
                         "convolutions": "engine (mvd_conv2d_split_f32: fp32 operands split into 2 fp16 terms, 3 products on fp16 MFMA, "
                                         "fp32 accumulate; channel-last, concat buffers written in place)",
                         "roofline": {"bound": "valu", "kernel": "sweep_corr (K1)", "achieved": k1_flops / (k1_ms * 1e-3) / 1e12,
                                      "peak": 157.3, "unit": "TFLOP/s", "frac": k1_flops / (k1_ms * 1e-3) / 1e12 / 157.3,
                                      "algorithmic_flops_per_launch": k1_flops, "algorithmic_bytes_per_launch": k1_bytes,
                                      "algorithmic_gbs": k1_bytes / (k1_ms * 1e-3) / 1e9, "avg_launch_ms": k1_ms,
                                      "launches_timed": args.steps, "pmc": k1_pmc}}
        # second roofline block of Path A: the longest convolution of its 2-D CNN (encoder conv2: 5x5, stride 2, 64 -> 128, the V source
        # views as one batch) on the split-operand implicit-GEMM kernel, against the dense fp16 matrix peak; timed standalone
        from robustmvd_amd import ops as ops_a
        wts2 = ma._engine._prepare()["conv2"]
        x2 = torch.rand(V, H // 2, W // 2, 64, device=dev)
        am2, ya2 = ops_a.absmax(x2), torch.zeros(1, device=dev)
        for _ in range(3):
            y2 = ops_a.conv2d_split(x2, am2, wts2, out_absmax=ya2)
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record()
        for _ in range(10):
            y2 = ops_a.conv2d_split(x2, am2, wts2, out_absmax=ya2)
        f1.record()
        torch.cuda.synchronize(dev)
        ms2 = f0.elapsed_time(f1) / 10
        px2 = V * (H // 4) * (W // 4)
        flops2 = 2.0 * px2 * 128 * 64 * 25                               # the layer's arithmetic
        issued2 = 3 * 2.0 * px2 * 128 * (64 // 8) * 28 * 8               # 3 products; per 8-channel chunk 7 K steps = 28 units of 8 (25 taps)
        out["path_a"]["roofline_mfma"] = {
            "bound": "mfma", "kernel": "conv2d_split_kernel<5,5,2> (encoder conv2, 64->128, 5x5 stride 2, %d views)" % V,
            "achieved": issued2 / ms2 / 1e9, "peak": 2500.0, "unit": "TFLOP/s", "frac": issued2 / ms2 / 1e9 / 2500.0,
            "issued_flops_per_launch": issued2, "algorithmic_flops_per_launch": flops2, "fp32_equivalent_tflops": flops2 / ms2 / 1e9,
            "avg_launch_ms": ms2, "launches_timed": 10,
            "note": "dense fp16 MFMA peak; fp32 operands as two fp16 terms, three products per multiply; matrix pipe 38 % busy by "
                    "counters (profiles/r03_conv2d_5x5_pmc.txt), the rest is patch staging and waits (DESIGN.md 4)"}
        del x2, y2
        # two frames in flight (FramePipeline: two HIP streams): the decoder's small layers of one frame overlap the encoder of the next
        import robustmvd_amd as R2
        pipe_a = R2.FramePipeline(ma, depth=2)

        def run_pipe_a(nfr):
            for i in range(nfr):
                pipe_a.submit(**sa[i % len(sa)])

        run_pipe_a(args.warmup + 2)
        torch.cuda.synchronize(dev)
        import torch.distributed as dist_a
        from robustmvd_amd.sharding import timed_region as timed_region_a
        dtpa = timed_region_a(lambda: run_pipe_a(args.steps), sync=lambda: torch.cuda.synchronize(dev),
                              dist=dist_a if world > 1 else None, device=cdev)
        out["path_a_pipelined"] = {"model": "robust_mvd", "frames_in_flight": 2, "value": world * args.steps / dtpa, "unit": "depth-maps/sec",
                                   "ms_per_step": dtpa / args.steps * 1e3, "note": "two HIP streams per process; not the path_a value"}
        del pipe_a
        # the same model with its 2-D CNN layer by layer on the vendor library's convolutions (engine_dispnet=False: the round-2 form)
        mv, _ = build_robustmvd(dev, engine_dispnet=False)
        with torch.no_grad():
            d0 = ma(**sa[0])[0]["depth"].clone()
            d1 = mv(**sa[0])[0]["depth"].clone()
        torch.cuda.synchronize(dev)
        dtv = timed_loop(mv, sa, args.steps, args.warmup, world, dev, None, cdev)
        out["path_a_vendor_convs"] = {"model": "robust_mvd(engine_dispnet=False)", "value": world * args.steps / dtv, "unit": "depth-maps/sec",
                                      "ms_per_step": dtv / args.steps * 1e3,
                                      "max_rel_depth_diff_vs_path_a": float(((d1 - d0).abs() / d0.abs()).max()),
                                      "note": "MIOpen / rocBLAS convolutions + fused bias/LeakyReLU pass; not the default.  The difference is relative to DEPTH "
                                              "(1 / inverse depth: far pixels amplify it); the inverse depths agree to 4e-6 absolute"}
        del ma, mv
        torch.cuda.empty_cache()
        # opt-in variant, NOT the fp32 drop-in: the DispNet's 2-D convolutions on the vendor library's fp16 kernels under
        # autocast (SURVEY.md 8f rank 1 lists fp16 as a tuning lever of that row); sweep, fusion and heads stay fp32
        mh, _ = build_robustmvd(dev, half_dispnet=True)
        dth = timed_loop(mh, sa, args.steps, args.warmup, world, dev, None, cdev)
        out["path_a_half_dispnet"] = {"model": "robust_mvd(half_dispnet=True)", "value": world * args.steps / dth,
                                      "unit": "depth-maps/sec", "ms_per_step": dth / args.steps * 1e3,
                                      "dtype": "f16 vendor convolutions (f32 accumulate) around an f32 sweep",
                                      "note": "opt-in; inverse depth within 2e-2 of the fp32 model (tests/test_hip_configs.py)"}
        del mh, sa
        torch.cuda.empty_cache()


    if extras and not args.no_path_a and args.config in (1, 2, 3):
        guarded("path_a", path_a_block)

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not stub:
        guarded("cpu_baseline", lambda: out.__setitem__("cpu_baseline", cpu_baseline(H, W, V, D, sd)))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
