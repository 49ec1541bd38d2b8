"""Benchmark of the hot path: ConQueR / Voxel-DETR single-frame Waymo TRAINING STEP on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A step = voxelize (GPU) -> sparse-conv backbone -> BEV/FPN -> box-attention DETR enc/dec -> losses ->
backward -> (gradient all-reduce over RCCL) -> AdamW, on `--scenes` synthetic Waymo-shaped scenes per GPU
that are resident in HBM before the timed region.  Prints ONE JSON line (rank 0).
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
    ap.add_argument("--scenes", type=int, default=2, help="scenes per GPU (configs[1]: batch 2; configs[2]: 16 / 8 GPUs)")
    ap.add_argument("--points", type=int, default=180000)
    ap.add_argument("--queries", type=int, default=1000, help="reference YAML default (configs[2] names 900)")
    ap.add_argument("--pool", type=int, default=2, help="distinct synthetic batches cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-graph", action="store_true", help="also evaluate the unused FPN levels like the reference")
    return ap.parse_args()


def _pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/pmc_latest.json,
    produced on the GPU box by scripts/round_profile.sh + scripts/pmc_to_json.py), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            kernels = json.load(f)["kernels"]
        if kernel not in kernels and "+" in kernel:  # "a+b": a label that times two kernels together
            parts = kernel.split("+")
            if all(k in kernels for k in parts):
                return sum(kernels[k]["hbm_bytes_per_launch"] for k in parts)
        return kernels[kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(args):
    """The same train step on the host cores with the ORACLE standing in for every HIP op
    (oracle/cpu_backend.py) -- a bounded sample: ONE config-0 sized scene (16k points)."""
    import oracle  # noqa: F401  (test/bench-only checker)
    from oracle import cpu_backend

    from efg_amd.engine import Trainer, synthetic_batch

    cores = min(os.cpu_count() or 1, 16)  # more threads only add oversubscription on this workload
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    tr = Trainer(device="cpu", overrides={"model.transformer.num_queries": args.queries}, seed=0, ddp=False)
    batch = synthetic_batch(1000, 1, n_points=16000)
    with cpu_backend.install():
        t0 = time.perf_counter()
        tr.step(batch)
        dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "scenes/s", "cores": cores, "kind": "port",
            "sample": "1 train step (fwd+bwd+AdamW) on 1 synthetic scene of 16k points (BASELINE config 0 cloud), "
                      "full ConQueR model (%d queries), oracle C ops (OpenMP) + PyTorch CPU dense layers, %.1f s" % (
                          args.queries, dt)}


def main():
    args = parse()
    from efg_amd import _prof
    from efg_amd.engine import Trainer, init_distributed, synthetic_batch

    rank, local_rank, world = init_distributed()
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    overrides = {"model.transformer.num_queries": args.queries}
    if args.full_graph:
        overrides["model.eval_unused_levels"] = True
    trainer = Trainer(device=dev, overrides=overrides, seed=0)
    # rank-sharded scenes: scene ids are disjoint across ranks (weak scaling: fixed per-GPU work)
    pool = [synthetic_batch(2000 + 100 * p + rank * args.scenes, args.scenes, n_points=args.points, device=dev)
            for p in range(args.pool)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup == 0:
        # lazy initialisation (code-object load, MIOpen find, TunableOp validation: ~8 s in the first step of a fresh
        # process, scripts/ubench/first_steps.py) is start-up, not a step; with W >= 1 the warm-up absorbs it
        trainer.step(pool[0])
    for w in range(args.warmup):
        trainer.step(pool[w % len(pool)])
    barrier()
    _prof.enable(True)
    t0 = time.perf_counter()
    for s in range(args.steps):
        trainer.step(pool[s % len(pool)])
    barrier()
    elapsed = time.perf_counter() - t0
    _prof.enable(False)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    scenes_total = args.scenes * world * args.steps
    line = {
        "metric": "scenes/sec ConQueR 1-frame Waymo train step",
        "value": scenes_total / elapsed,
        "unit": "scenes/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "ConQueR/Voxel-DETR res18 p3, 1-frame Waymo-shaped scenes, %d pts/scene, 0.1 m voxels, "
                        "%d scenes/GPU, %d queries, fwd+bwd+AdamW%s" % (args.points, args.scenes, args.queries,
                                                                     ", full reference graph" if args.full_graph
                                                                     else ", unused FPN levels not evaluated"),
            "global_batch": args.scenes * world,
            "parallelism": "dp%d" % world,
        },
    }
    if rank == 0:
        roof = _prof.roofline()
        if roof is not None:
            roof["traffic"] = _pmc_traffic(roof["kernel"])
        line["roofline"] = roof
        line["kernels"] = {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 1),
                               "GBps_alg": round(v["bytes"] / max(v["total_ms"], 1e-9) / 1e6, 1),
                               "TFLOPs_alg": round(v["flops"] / max(v["total_ms"], 1e-9) / 1e9, 2)}
                           for k, v in sorted(_prof.summary().items())}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
