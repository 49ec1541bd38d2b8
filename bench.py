"""Benchmark of the hot path: ConQueR / Voxel-DETR single-frame Waymo TRAINING STEP on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A step = voxelize (GPU) -> sparse-conv backbone -> BEV/FPN -> box-attention DETR enc/dec -> 32 losses (device
Hungarian matching) -> backward -> (gradient all-reduce over RCCL) -> AdamW + OneCycle schedule, on `--scenes`
synthetic Waymo-shaped scenes per GPU that are resident in HBM before the timed region.  Prints ONE JSON line (rank 0).

The timed region is clean: no event records, no count read-backs for accounting.  The per-kernel numbers (`roofline`,
`kernels`, `geometry`) come from `--profile-steps` EXTRA steps after the timed region, with HIP events on the launch
stream around every launch of our kernels; `full_graph` is a second, shorter timing of the same step with the FPN
levels / heads the reference evaluates and never reads; `cpu_baseline` is the reference / oracle on the
host cores (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from efg_amd.engine import configure_hip_runtime  # noqa: E402

configure_hip_runtime()  # GPU_MAX_HW_QUEUES, before the first HIP call of the process


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="conquer", choices=["conquer", "voxeldetr", "centerpoint", "trajectoryformer"],
                    help="conquer = BASELINE configs[1]/[2] (default); voxeldetr = the plain variant; centerpoint = "
                         "configs[0]/[3] (VoxelNet: reader -> SpMiddleResNetFHD -> RPN -> CenterHead)")
    ap.add_argument("--scenes", type=int, default=2, help="scenes per GPU (configs[1]: batch 2; configs[2]: 16 / 8 GPUs)")
    ap.add_argument("--points", type=int, default=180000)
    ap.add_argument("--dense", action="store_true",
                    help="dense scenes: 55 %% isolated returns => ~140k occupied voxels per 180k-point sweep, i.e. the "
                         "120 000-voxel training cap and its break semantics are hit in the timed step")
    ap.add_argument("--objects", type=int, default=60, help="trajectoryformer: annotated objects per sample")
    ap.add_argument("--sweeps", type=int, default=1, help="4 = the 720k-point multi-sweep cloud of configs[3] (6 features)")
    ap.add_argument("--queries", type=int, default=900,
                    help="BASELINE.json configs[2]: 900 queries (the reference YAML's own default is 1000: the line's "
                         "`yaml_config` object times that)")
    ap.add_argument("--pool", type=int, default=2, help="distinct synthetic batches cycled through")
    ap.add_argument("--profile-steps", type=int, default=3, help="extra, untimed steps with per-kernel HIP events")
    ap.add_argument("--soak-steps", type=int, default=300,
                    help="N > 0: after the timed run, N more optimizer steps on a 4-batch pool, then the step timed again -> the "
                         "line's `steady_state` object (the encoder's boxes grow as training proceeds; the default is a ~10 s "
                         "soak, 0 skips it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-arm", action="store_true", help="skip the third timing (bf16x3 split-precision A/B arm)")
    ap.add_argument("--no-full-graph", action="store_true", help="skip the second timing with the unused FPN levels")
    ap.add_argument("--full-graph", action="store_true", help="make the full reference graph the headline run")
    ap.add_argument("--ddp-mode", default=None, choices=["bucket", "flat", "static", "find_unused", "plain"],
                    help="N > 1 gradient exchange (engine.Trainer): flat = one all-reduce after backward (default); bucket = "
                         "three flat buckets all-reduced on a communication stream while backward still runs (what the "
                         "reference's DDP does; not the default, see engine.py); static / find_unused / plain = torch "
                         "DistributedDataParallel variants")
    return ap.parse_args()


def _pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from this round's committed rocprofv3 --pmc passes over THIS command
    (profiles/pmc_latest.json, written on the GPU box by scripts/round_profile.sh + scripts/pmc_to_json.py; counters
    cannot be read from inside the process), or None when the file has no entry for the kernel."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            doc = json.load(f)
        kernels = doc["kernels"]
        if kernel not in kernels and "+" in kernel:  # "a+b": a label that times two kernels together
            parts = kernel.split("+")
            if all(k in kernels for k in parts):
                return sum(kernels[k]["hbm_bytes_per_launch"] for k in parts), doc.get("source")
        return kernels[kernel]["hbm_bytes_per_launch"], doc.get("source")
    except Exception:
        return None, None


def _pmc_prefix(prefix):
    """(sum of HBM bytes per launch, names) over the kernels of profiles/pmc_latest.json whose name starts with `prefix`
    (one launch of each per call), or (None, None)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            kernels = json.load(f)["kernels"]
        hit = {k: v["hbm_bytes_per_launch"] for k, v in kernels.items() if k.startswith(prefix)}
        return (sum(hit.values()), sorted(hit)) if hit else (None, None)
    except Exception:
        return None, None


def cpu_baseline(args):
    """Host-core baseline on the GPU box (bounded: ~20-30 s), same scene size as the GPU workload:
      * value: the whole train step (fwd + bwd + AdamW) on ONE `--points`-point scene with the ORACLE standing in for
        every HIP op (oracle/cpu_backend.py, OpenMP C) and PyTorch CPU dense layers -- kind "port";
      * stages: the reference's own voxelizer (oracle/_ref = voxelization_cpu.cpp compiled in place, 1 thread, the
        loop is serial) when it was built, the oracle voxelizer, and the oracle sparse backbone forward."""
    import numpy as np

    import oracle  # noqa: F401  (test/bench-only checker)
    from oracle import cpu_backend

    from efg_amd.data.synthetic import PC_RANGE, VOXEL_SIZE, make_scene
    from efg_amd.engine import Trainer, synthetic_batch

    cores = min(os.cpu_count() or 1, 16)  # more threads only add oversubscription on this workload
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    stages = {}
    pts = make_scene(1000, n_points=args.points, n_sweeps=args.sweeps)[0]
    cap = 120000 if args.sweeps == 1 else 200000
    t0 = time.perf_counter()
    v, c, n = oracle.hard_voxelize(pts, VOXEL_SIZE, PC_RANGE, 5, cap)
    stages["hard_voxelize oracle (C, 1 thread) s/scene"] = round(time.perf_counter() - t0, 4)
    if oracle.ref_available():
        t0 = time.perf_counter()
        oracle.hard_voxelize(pts, VOXEL_SIZE, PC_RANGE, 5, cap, use_ref=True)
        stages["hard_voxelize REFERENCE voxelization_cpu.cpp (1 thread) s/scene"] = round(time.perf_counter() - t0, 4)
    stages["voxels/scene"] = int(v.shape[0])
    tr = Trainer(device="cpu", overrides={"model.transformer.num_queries": args.queries}, seed=0, ddp=False)
    tr.model.noise_generator = torch.Generator().manual_seed(4321)   # (the CDN noise of the parity check below)
    batch = synthetic_batch(1000, 1, n_points=args.points, n_sweeps=args.sweeps)
    with cpu_backend.install():
        t0 = time.perf_counter()
        with torch.no_grad():
            tr.model.backbone.extractor.bottom_up(torch.from_numpy(oracle.voxel_mean(v, n)),
                                                  torch.from_numpy(np.pad(c, ((0, 0), (1, 0)))), 1, [1504, 1504, 40])
        stages["sparse backbone forward (oracle C, OpenMP) s/scene"] = round(time.perf_counter() - t0, 3)
        t0 = time.perf_counter()
        cpu_losses, _ = tr.step(batch)
        dt = time.perf_counter() - t0
    cpu_losses = {k: float(v.detach()) for k, v in cpu_losses.items()}
    cpu_losses["__grad_norm__"] = _grad_norm(tr.model)
    _LAST_CPU_LOSSES.clear()
    _LAST_CPU_LOSSES.update(cpu_losses)
    tr.close()
    cpu = "?"
    try:
        with open("/proc/cpuinfo") as f:
            cpu = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")][0]
    except Exception:
        pass
    return {"parity_full_size": _parity_full_size(args, cpu_losses),
            "value": 1.0 / dt, "unit": "scenes/s", "cores": cores, "kind": "port", "cpu": cpu,
            "sample": "1 train step (fwd+bwd+AdamW) on 1 synthetic scene of %d points (the GPU workload's scene size), "
                      "full ConQueR model (%d queries), oracle C ops (OpenMP) + PyTorch CPU dense layers, %.1f s"
                      % (args.points, args.queries, dt),
            "stages": stages}


_LAST_CPU_LOSSES = {}   # the oracle-backed CPU step of cpu_baseline(), kept for the arm's parity check


def _grad_norm(model):
    """l2 norm of all gradients (fp64 accumulation) after a step."""
    return float(torch.sqrt(sum((p.grad.double() ** 2).sum().cpu() for p in model.parameters() if p.grad is not None)))


def _parity_full_size(args, cpu_losses):
    """The SAME step (same seed-0 weights, same 180k-point scene, same CDN noise) on the HIP path: every loss term of
    the GPU step against the oracle-backed CPU step that was just timed -- a parity check at the benchmark's full
    scene size, which the oracle-sized tests cannot reach (the CPU side takes ~9 s)."""
    from efg_amd.engine import Trainer, synthetic_batch

    dev = torch.device("cuda", torch.cuda.current_device())
    tr = Trainer(device=dev, overrides={"model.transformer.num_queries": args.queries}, seed=0, ddp=False)
    tr.model.noise_generator = torch.Generator().manual_seed(4321)
    gpu_losses, _ = tr.step(synthetic_batch(1000, 1, n_points=args.points, n_sweeps=args.sweeps, device=dev))
    gpu_losses = {k: float(v.detach()) for k, v in gpu_losses.items()}
    gpu_norm = _grad_norm(tr.model)
    tr.close()
    terms = sorted(k for k in cpu_losses if k.startswith("loss"))
    rel = {k: abs(gpu_losses[k] - cpu_losses[k]) / max(abs(cpu_losses[k]), 1e-6) for k in terms}
    worst = max(rel, key=rel.get)
    return {"what": "every loss term of one full-size train step, HIP path vs the oracle-backed CPU step (same weights, "
                    "scene, CDN noise)", "terms": len(terms), "total_cpu": sum(cpu_losses[k] for k in terms),
            "total_gpu": sum(gpu_losses[k] for k in terms), "max_rel_diff": rel[worst], "worst_term": worst,
            "grad_norm_cpu": cpu_losses["__grad_norm__"], "grad_norm_gpu": gpu_norm,
            "grad_norm_rel_diff": abs(gpu_norm - cpu_losses["__grad_norm__"]) / max(cpu_losses["__grad_norm__"], 1e-12)}


def _voxelize_alone(trainer, batch, iters=30):
    """(microseconds per call on the DEVICE, microseconds per call as the step issues it, algorithmic bytes per call) of the
    rank's batched hard voxelization with nothing else running.  The device figure is the MEDIAN over the iterations of the
    time between two events recorded around ONE call's launches (the C entry point queues its memset + 6 kernels in a few
    tens of microseconds, faster than they run, so the events bracket back-to-back device work whatever the host does
    between calls); the second figure is the op's own entry point in a loop, each call ending in its voxel-count read-back
    (a host round trip per iteration: 50+ us on a loaded box on top of 80 us of kernels).  (A HIP-graph replay of the call
    faulted on this stack -- "write access to a read-only page" -- and is not used.)"""
    from efg_amd.operators import voxelize as V
    from efg_amd.operators import voxelize_batch

    cfg = trainer.cfg.dataset
    vox = cfg.processors.train.Voxelization
    pts = [b[0]["points"] for b in batch]
    args = (list(cfg.voxel_size), list(cfg.pc_range), vox.max_points_in_voxel, vox.max_voxel_num)
    run = lambda: voxelize_batch(pts, *args)  # noqa: E731
    out = run()
    torch.cuda.synchronize()
    n, f = sum(p.shape[0] for p in pts), pts[0].shape[1]
    m = int(out["voxels"].shape[0])
    nbytes = 4 * f * n + m * (4 * vox.max_points_in_voxel * f + 16 + 4 + 4 * f)

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters

    call_us = timed(run)
    dev_us = None
    try:
        points = torch.cat(pts, 0).contiguous()
        offsets = [0]
        for p_ in pts:
            offsets.append(offsets[-1] + p_.shape[0])
        cap = min(len(pts) * vox.max_voxel_num, n)
        bufs = (torch.empty((cap, vox.max_points_in_voxel, f), dtype=torch.float32, device=points.device),
                torch.empty((cap, 4), dtype=torch.int32, device=points.device),
                torch.empty((cap,), dtype=torch.int32, device=points.device),
                torch.zeros(len(pts), dtype=torch.int32, device=points.device),
                torch.empty((cap, f), dtype=torch.float32, device=points.device))
        pairs = []
        for _ in range(iters):
            bufs[3].zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            V._hard_voxelize_launch(points, offsets, args[0], args[1], args[2], args[3], bufs[0], bufs[1], bufs[2], bufs[3],
                                    bufs[4])
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        assert int(bufs[3].sum()) == m, "the raw voxelizer launches gave another voxel count"
        per_call = sorted(a_.elapsed_time(b_) * 1e3 for a_, b_ in pairs)
        dev_us = per_call[len(per_call) // 2]
    except Exception:  # noqa: BLE001 -- accounting only: keep the call-by-call figure
        dev_us = None
    return (dev_us if dev_us is not None else call_us), call_us, nbytes


def _geometry_report(trainer, batch):
    """N, M per level and pairs per sparse-conv table of one batch (SURVEY.md §8d asks for them beside every GB/s
    figure).  One untimed forward of the backbone with a spy on the rulebook constructor."""
    import efg_amd.spconv.core as core

    seen = []
    orig = core.Rulebook.__init__

    def spy(self, *a, **k):
        orig(self, *a, **k)
        seen.append(self)

    core.Rulebook.__init__ = spy
    try:
        model = trainer.model
        with torch.no_grad():
            voxels, coords, npv, shape, mean = model._inputs(batch)
            model.backbone(voxels, coords, npv, len(batch), shape, mean)
    finally:
        core.Rulebook.__init__ = orig
    torch.cuda.synchronize()
    n_points = int(sum(b[0]["points"].shape[0] for b in batch))
    tables = [{"kind": "subm" if rb.subm else "strided", "kvol": rb.kvol, "m_in": rb.m_in, "m_out": rb.m_out,
               "pairs": rb.num_pairs()} for rb in seen]
    return {"points": n_points, "input_voxels": int(coords.shape[0]), "tables": tables}


def _rank_report(trainer, batch, world, dev):
    """Per-rank facts gathered after the timed region: input voxels of the rank's first batch (scenes differ in size: this
    is the weak-scaling imbalance) and the number of stream-K units that had to be recomputed because a share did not
    arrive within the bounded wait (csrc/spconv_tiles.hip; 0 in a healthy run -- a non-zero count means a silent 2x on
    those units, e.g. two processes spinning on one device)."""
    import ctypes

    from efg_amd import _lib
    from efg_amd.operators import voxelize_batch

    cfg = trainer.cfg.dataset
    vox = cfg.processors.train.Voxelization
    out = voxelize_batch([b[0]["points"] for b in batch], list(cfg.voxel_size), list(cfg.pc_range),
                         vox.max_points_in_voxel, vox.max_voxel_num, with_mean=False)
    n = ctypes.c_int64(0)
    _lib.check(_lib.lib().efg_spconv_streamk_fallbacks(ctypes.byref(n), 0))
    mine = torch.tensor([sum(out["num_voxels"]), n.value], device=dev, dtype=torch.int64)
    every = [mine]
    if world > 1:
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
    return {"rank_input_voxels": [int(t[0]) for t in every], "streamk_fallbacks": [int(t[1]) for t in every]}


def _pin_rank(local_rank, world):
    """N > 1: every rank's threads (Python main + autograd thread, ROCr's event thread: ~2 busy cores per rank, DESIGN.md
    section 8) on a slice of the CPUs this process may use that no other rank of the node gets -- eight ranks otherwise
    migrate over each other's cores.  A slice, not two cores: the scheduler keeps its freedom inside it (the boxes of this
    pool are shared; pinning to fixed cores loses to whatever else runs there).  EFG_PIN_RANKS=0: off.  Returns the slice."""
    if world <= 1 or os.environ.get("EFG_PIN_RANKS", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = sorted(os.sched_getaffinity(0))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    per = len(cpus) // max(local_world, 1)
    if per < 2:
        return None
    mine = cpus[(local_rank % local_world) * per:(local_rank % local_world + 1) * per]
    try:
        for tid in os.listdir("/proc/self/task"):   # threads that exist already (the HIP runtime's); later ones inherit
            try:
                os.sched_setaffinity(int(tid), mine)
            except OSError:
                pass
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return [mine[0], mine[-1]]


def _host_own_ms(trainer, pool, dev, steps=4):
    """The host's OWN time to queue one step: issue time measured from EMPTY device queues (in steady state the launches
    block on the full queue and `host_issue_ms_per_step` reads step - ~2 ms whatever the host needs).  Includes the blocking
    site-count read-backs of the geometry stream, which wait for ~1 ms of device work."""
    tot = 0.0
    for s in range(steps):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        trainer.step(pool[s % len(pool)])
        tot += time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    return 1000.0 * tot / steps


def _self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command under torch.distributed.run with one rank per
    GPU (what the reference's efg/engine/launch.py:52-57 does with mp.spawn), rendezvous on 127.0.0.1, rank 0 prints the
    one JSON line.  Fails up front, with the reason, when the node has fewer than N devices (EFG_DIST_BACKEND=gloo lets
    ranks share a device: the one-GPU test of the N > 1 path)."""
    import socket
    import subprocess

    have = torch.cuda.device_count()
    if have < n and os.environ.get("EFG_DIST_BACKEND", "nccl") != "gloo":
        raise SystemExit("bench.py: --gpus %d needs %d devices, this node has %d (RCCL cannot share a device between "
                         "ranks)" % (n, n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    from efg_amd import _prof
    from efg_amd.engine import Trainer, init_distributed, synthetic_batch

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return _self_launch(args.gpus)   # bare `python bench.py --gpus N`: start the N ranks ourselves
    rank, local_rank, world = init_distributed()
    pinned = _pin_rank(local_rank, world)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, "
                         "or run `python bench.py --gpus %d` bare and it starts the ranks itself)"
                         % (args.gpus, world, args.gpus, args.gpus))
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    if args.model == "trajectoryformer":
        from efg_amd.tracking.bench import run as run_tracking

        return run_tracking(args, rank, local_rank, world, dev)
    if args.model == "centerpoint":
        from efg_amd.centerpoint.bench import run as run_centerpoint

        return run_centerpoint(args, rank, local_rank, world, dev)
    config = None if args.model == "conquer" else os.path.join(ROOT, "configs", "voxeldetr_waymo_res18.yaml")
    overrides = {"model.transformer.num_queries": args.queries}
    if args.full_graph:
        overrides["model.eval_unused_levels"] = True
    trainer = Trainer(config=config, device=dev, overrides=overrides, seed=0, ddp_mode=args.ddp_mode)
    # rank-sharded scenes: scene ids are disjoint across ranks (weak scaling: fixed per-GPU work)
    pool = [synthetic_batch(2000 + 100 * p + rank * args.scenes, args.scenes, n_points=args.points, device=dev,
                            n_sweeps=args.sweeps, clutter=0.55 if args.dense else 0.0) for p in range(args.pool)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    stats = {}

    def timed_run(tr, steps, warmup):
        if warmup == 0:
            # lazy initialisation (code-object load, MIOpen find, TunableOp validation: ~8 s in the first step of a
            # fresh process, scripts/ubench/first_steps.py) is start-up, not a step; with W >= 1 the warm-up absorbs it
            tr.step(pool[0])
        for w in range(warmup):
            tr.step(pool[w % len(pool)])
        barrier()
        t0 = time.perf_counter()
        for s in range(steps):
            tr.step(pool[s % len(pool)])
        issued = time.perf_counter() - t0     # the host has QUEUED the last step here; the device may still be running
        barrier()
        elapsed = time.perf_counter() - t0
        stats["host_issue_ms_per_step"] = 1000.0 * issued / steps
        stats["rank_ms_per_step"] = [1000.0 * elapsed / steps]
        if world > 1:
            mine = torch.tensor([elapsed, issued], device=dev, dtype=torch.float64)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            stats["rank_ms_per_step"] = [1000.0 * float(t[0]) / steps for t in every]
            stats["host_issue_ms_per_step"] = max(1000.0 * float(t[1]) / steps for t in every)
            elapsed = max(float(t[0]) for t in every)     # MAX over ranks
        return elapsed

    elapsed = timed_run(trainer, args.steps, args.warmup)
    scenes_total = args.scenes * world * args.steps
    graph = ("full reference graph" if args.full_graph else
             "FPN levels / heads the reference evaluates and never reads are skipped (same losses and gradients; the line's "
             "`full_graph` object times the full reference graph)")
    line = {
        "metric": "scenes/sec %s 1-frame Waymo train step" % {"conquer": "ConQueR", "voxeldetr": "Voxel-DETR"}[args.model],
        "value": scenes_total / elapsed,
        "unit": "scenes/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        # EFG_GEMM_ARM=bf16x3 is the A/B arm (split-precision products in the long Linear layers, csrc/gemm_bf16x3.hip): it
        # reports under its own label and is never the default
        "dtype": "f32 + bf16x3 (split bf16 products, fp32 accumulate, in the encoder-sized Linear forward / data-gradient)"
                 if os.environ.get("EFG_GEMM_ARM", "") == "bf16x3" else "f32",
        "data": "synthetic",
        "config": {
            "workload": "%s res18 p3, %d-sweep Waymo-shaped scenes%s, %d pts/scene, 0.1 m voxels, %d scenes/GPU, %d queries, "
                        "fwd+bwd+AdamW+OneCycle, graph of `value`: %s" % ({"conquer": "ConQueR", "voxeldetr": "Voxel-DETR"}[args.model],
                                                        args.sweeps, " (dense preset: voxel cap hit)" if args.dense else "",
                                                        args.points, args.scenes, args.queries, graph),
            "global_batch": args.scenes * world,
            "parallelism": "dp%d" % world,
        },
        # what the collective library saw (the driver can check that RCCL ran with N ranks)
        "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
        "dist_backend": ("rccl" if dist.get_backend() == "nccl" else dist.get_backend()) if dist.is_initialized() else None,
        # N > 1 readiness: which exchange ran, every rank's own step time (the spread is the load imbalance between the
        # ranks' scenes plus host jitter), and the time the slowest HOST needed to queue a step.  NOTE: with the device a step
        # behind, launches block on the full queue, so in steady state this reads (step - ~2 ms) on a device-bound box too; the
        # host's own work is 20-24 ms per step (profiles/r05c_geom_prefetch.txt).  Only a value ABOVE the kernels' time per step
        # means a host-bound step.
        "ddp_mode": trainer.ddp_mode,
        "rank_ms_per_step": {"min": round(min(stats["rank_ms_per_step"]), 3), "max": round(max(stats["rank_ms_per_step"]), 3),
                             "all": [round(t, 3) for t in stats["rank_ms_per_step"]]},
        "host_issue_ms_per_step": round(stats["host_issue_ms_per_step"], 3),
    }
    line.update(_rank_report(trainer, pool[0], world, dev))
    own = torch.tensor([_host_own_ms(trainer, pool, dev)], device=dev, dtype=torch.float64)
    every = [own]
    if world > 1:
        every = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(every, own)
    line["host_own_ms_per_step"] = [round(float(t[0]), 3) for t in every]   # per rank, from empty queues (see _host_own_ms)
    line["rank_cpu_slice"] = pinned                                           # rank 0's [first, last] CPU, None = not pinned
    line["hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES")
    # ---- per-kernel numbers: extra steps, outside the timed region ------------------------------------------------
    if args.profile_steps > 0:
        _prof.enable(True)
        for s in range(args.profile_steps):
            trainer.step(pool[s % len(pool)])
        barrier()
        _prof.enable(False)
    if rank == 0:
        summ = _prof.summary() if args.profile_steps > 0 else {}
        roof = _prof.roofline() if summ else None
        if roof is not None:
            roof["traffic"], roof["traffic_source"] = _pmc_traffic(roof["kernel"])
            roof["measured"] = "HIP events on the launch stream, %d extra steps after the timed region" % args.profile_steps
        line["roofline"] = roof
        line["kernels"] = {k: {"launches_per_step": round(v["launches"] / max(args.profile_steps, 1), 1),
                               "avg_us": round(v["avg_us"], 1),
                               "GBps_alg": round(v["bytes"] / max(v["total_ms"], 1e-9) / 1e6, 1),
                               "TFLOPs_alg": round(v["flops"] / max(v["total_ms"], 1e-9) / 1e9, 2)}
                           for k, v in sorted(summ.items())}
        # BASELINE's second metric: voxelize + spconv algorithmic HBM GB/s (fraction of the 8 TB/s roof)
        vox = {k: v for k, v in summ.items() if k.startswith("hard_voxelize")}
        conv = {k: v for k, v in summ.items() if k.startswith("conv_")}
        for name, grp in (("voxelize", vox), ("spconv", conv)):
            ms = sum(v["total_ms"] for v in grp.values())
            if ms > 0:
                gbs = sum(v["bytes"] for v in grp.values()) / ms / 1e6
                line[name + "_hbm"] = {"GBps_alg": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000.0, 4),
                                       "ms_per_step": round(ms / max(args.profile_steps, 1), 3)}
        if "voxelize_hbm" in line:
            # the same call alone on an otherwise idle GPU (inside the step it runs on the geometry stream beside the
            # previous step's backward, so its in-step duration mostly measures contention)
            try:
                us, call_us, nbytes = _voxelize_alone(trainer, pool[0])
                line["voxelize_hbm"].update({"standalone_us": round(us, 1), "standalone_call_us": round(call_us, 1),
                                             "standalone_GBps_alg": round(nbytes / us / 1e3, 1),
                                             "standalone_frac_of_8TBps": round(nbytes / us / 1e3 / 8000.0, 4),
                                             "standalone_note": "standalone_us: median time between two events around ONE call's "
                                             "launches = device time; standalone_call_us: the op called in a loop, each call "
                                             "ending in its voxel-count read-back"})
            except Exception as exc:  # accounting only
                line["voxelize_hbm"]["standalone_error"] = str(exc)
            # counter traffic of the call's kernels (every `vox_*` kernel of profiles/pmc_latest.json: the same PMC passes as
            # roofline.traffic; the one memset of the call -- ~1 MB of counters -- is not an efg kernel and is left out)
            tr_b, names = _pmc_prefix("vox_")
            if tr_b is not None:
                alg = sum(v["bytes"] for v in vox.values()) / max(sum(v["launches"] for v in vox.values()), 1)
                line["voxelize_hbm"].update({"traffic_bytes_per_call": round(tr_b), "alg_bytes_per_call": round(alg),
                                             "traffic_ratio": round(tr_b / max(alg, 1.0), 2), "traffic_kernels": names})
        try:
            line["geometry"] = _geometry_report(trainer, pool[0])
        except Exception as exc:  # accounting only
            line["geometry"] = {"error": str(exc)}
    # ---- A/B arm: the same step with split-precision (bf16 x 3) products in the encoder-sized Linear layers -----------
    # On the SAME trainer (the switch is read at call time): same memory, streams and graphs as the fp32 timing above, so
    # the difference is the products and nothing else (a second model in the process draws new physical memory and, before
    # efg_amd/streams.py, new hardware queues: one model in three came out 2-3 ms slower).
    arm_on = os.environ.get("EFG_GEMM_ARM", "") == "bf16x3"
    if not args.no_arm and not arm_on and world == 1 and args.model in ("conquer", "voxeldetr"):
        import efg_amd.operators.linear as _lin

        _lin._ARM_BF16X3 = True
        try:
            e3 = timed_run(trainer, args.steps, args.warmup)
            line["arm_bf16x3"] = {
                "ms_per_step": 1000.0 * e3 / args.steps, "value": args.scenes * args.steps / e3, "unit": "scenes/s",
                "steps": args.steps, "warmup": args.warmup,
                "dtype": "f32 + bf16x3: forward, data-gradient and weight-gradient products of the >= 16384-row Linear layers, the "
                         "64-channel-wide sparse-conv tile kernel (EFG_CONV_ARM) and, since round 4, the neck's dense 3 x 3 "
                         "convolution (EFG_CONV2D_ARM; three products per pass over overlapping-row views of the padded map) as "
                         "hi.hi + hi.lo + lo.hi of bf16-split operands, fp32 accumulate (csrc/gemm_bf16x3.hip, csrc/spconv_tiles.hip); "
                         "everything else exact fp32",
                "note": "A/B arm, not the headline: `value` above is the exact-fp32 step"}
        finally:
            _lin._ARM_BF16X3 = False
    trainer.close()
    # ---- the step after N optimizer steps of a COMPLETE schedule (profiles/r05_soak.txt: what sustained training runs at) ----------
    # A fresh model's encoder boxes fit the box-attention backward's query-tile window; trained ones partly do not (those
    # corners take the binned path).  The timed trainer above is at the start of the config's long schedule, where nothing
    # moves yet: this one gets a one-cycle schedule of exactly N + steps iterations (as scripts/ubench/soak.py does), so the
    # weights move as far in N steps as they do over a training run.
    if args.soak_steps > 0 and world == 1:
        del trainer
        torch.cuda.empty_cache()
        soak_tr = Trainer(config=config, device=dev, overrides=overrides, seed=0, max_iters=args.soak_steps + 2 * args.steps + args.warmup + 10)
        soak_pool = pool + [synthetic_batch(2000 + 100 * p, args.scenes, n_points=args.points, device=dev, n_sweeps=args.sweeps,
                                            clutter=0.55 if args.dense else 0.0) for p in range(len(pool), 4)]
        saved_pool = list(pool)
        pool[:] = soak_pool
        try:
            e0 = timed_run(soak_tr, args.steps, args.warmup)
            for i in range(args.soak_steps):
                soak_tr.step(soak_pool[i % len(soak_pool)])
            e5 = timed_run(soak_tr, args.steps, 0)
        finally:
            pool[:] = saved_pool
        soak_tr.close()
        del soak_tr
        line["steady_state"] = {"ms_per_step": 1000.0 * e5 / args.steps, "value": args.scenes * args.steps / e5, "unit": "scenes/s",
                                "steps": args.steps, "after_optimizer_steps": args.soak_steps + args.steps + args.warmup,
                                "same_trainer_fresh_ms_per_step": 1000.0 * e0 / args.steps,
                                "host_issue_ms_per_step": round(stats["host_issue_ms_per_step"], 3),
                                "note": "a second trainer on a one-cycle schedule of exactly that many iterations and a pool of 4 "
                                        "synthetic batches: timed fresh, run --soak-steps optimizer steps, timed again"}
        trainer = None
    # ---- the same step with the reference's dead branches evaluated ----------------------------------
    if not args.no_full_graph and not args.full_graph and world == 1:
        del trainer
        torch.cuda.empty_cache()
        ov = dict(overrides)
        ov["model.eval_unused_levels"] = True
        full = Trainer(config=config, device=dev, overrides=ov, seed=0)
        steps = max(5, args.steps // 2)
        e2 = timed_run(full, steps, 3)
        full.close()
        line["full_graph"] = {"ms_per_step": 1000.0 * e2 / steps, "value": args.scenes * steps / e2, "steps": steps,
                              "note": "also evaluates FPN p2 / p4-output / p5 and res2_out like the reference; "
                                      "same losses and gradients"}
    # ---- `value` is BASELINE.json configs[2] (900 queries); the reference YAML's own default is 1000 ------------------------
    if args.queries != 1000 and world == 1 and not args.no_full_graph and args.model == "conquer":
        torch.cuda.empty_cache()
        ov = dict(overrides)
        ov["model.transformer.num_queries"] = 1000
        y1000 = Trainer(config=config, device=dev, overrides=ov, seed=0, ddp_mode=args.ddp_mode)
        steps = max(5, args.steps // 2)
        e9 = timed_run(y1000, steps, 3)
        y1000.close()
        del y1000
        line["yaml_config"] = {"ms_per_step": 1000.0 * e9 / steps, "value": args.scenes * steps / e9, "unit": "scenes/s",
                               "steps": steps, "queries": 1000,
                               "note": "the same step with the reference YAML's 1000 queries; `value` is BASELINE.json "
                                       "configs[2]'s 900 (per-GPU share of the batch-16 / 8-GPU config: 2 scenes)"}
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            base = cpu_baseline(args)
            line["parity_full_size"] = base.pop("parity_full_size")
            line["cpu_baseline"] = base
            if "arm_bf16x3" in line:   # the arm's own gate: the same full-size step against the same oracle-backed CPU step
                import efg_amd.operators.linear as _lin

                _lin._ARM_BF16X3 = True
                try:
                    line["arm_bf16x3"]["parity_full_size"] = _parity_full_size(args, _LAST_CPU_LOSSES)
                finally:
                    _lin._ARM_BF16X3 = False
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
