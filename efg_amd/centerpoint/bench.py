"""`bench.py --model centerpoint`: one CenterPoint training step (BASELINE configs[0] / [3]) -- voxelize (GPU) ->
SpMiddleResNetFHD (HIP sparse convs) -> RPN -> CenterHead -> focal + L1 losses -> backward -> gradient exchange ->
AdamW + OneCycle + gradient clipping (the reference's solver for this experiment).  `--sweeps 4 --points 720000
--scenes 1` is the multi-sweep voxelization stress case (6 point features, 200 000-voxel cap)."""
import json
import os
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(args, rank, local_rank, world, dev):
    from .. import _prof
    from ..engine import Trainer, synthetic_batch
    from .voxelnet import VoxelNet

    ov = {}
    if args.sweeps > 1:  # ...36e.4f.improved/config.yaml:11,44,53,64-67
        ov.update({"dataset.nsweeps": args.sweeps, "model.reader.num_input_features": 6,
                   "model.backbone.num_input_features": 6,
                   "dataset.processors.train.Voxelization.max_voxel_num": 200000,
                   "dataset.processors.val.Voxelization.max_voxel_num": 400000})
    trainer = Trainer(config=os.path.join(ROOT, "configs", "centerpoint_waymo_voxelnet.yaml"), device=dev, overrides=ov,
                      seed=0, model_cls=VoxelNet, max_iters=36 * (158081 // (6 * 8)))
    pool = [synthetic_batch(3000 + 100 * p + rank * args.scenes, args.scenes, n_points=args.points, device=dev,
                            n_sweeps=args.sweeps) for p in range(args.pool)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup == 0:
        trainer.step(pool[0])
    for w in range(args.warmup):
        trainer.step(pool[w % len(pool)])
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        trainer.step(pool[s % len(pool)])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    line = {
        "metric": "scenes/sec CenterPoint %d-frame Waymo train step" % args.sweeps,
        "value": args.scenes * world * args.steps / elapsed, "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "CenterPoint VoxelNet (SpMiddleResNetFHD + RPN + CenterHead), %d-sweep Waymo-shaped scenes, "
                               "%d pts/scene x %d features, 0.1 m voxels, %d scenes/GPU, fwd+bwd+AdamW+OneCycle+clip"
                               % (args.sweeps, args.points, 5 if args.sweeps == 1 else 6, args.scenes),
                   "global_batch": args.scenes * world, "parallelism": "dp%d" % world},
    }
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
            roof["measured"] = "HIP events on the launch stream, %d extra steps after the timed region" % args.profile_steps
        line["roofline"] = roof
        line["kernels"] = {k: {"launches_per_step": round(v["launches"] / max(args.profile_steps, 1), 1),
                               "avg_us": round(v["avg_us"], 1),
                               "GBps_alg": round(v["bytes"] / max(v["total_ms"], 1e-9) / 1e6, 1),
                               "TFLOPs_alg": round(v["flops"] / max(v["total_ms"], 1e-9) / 1e9, 2)}
                           for k, v in sorted(summ.items())}
        for name, prefix in (("voxelize", "hard_voxelize"), ("spconv", "conv_")):
            grp = {k: v for k, v in summ.items() if k.startswith(prefix)}
            ms = sum(v["total_ms"] for v in grp.values())
            if ms > 0:
                gbs = sum(v["bytes"] for v in grp.values()) / ms / 1e6
                line[name + "_hbm"] = {"GBps_alg": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000.0, 4),
                                       "ms_per_step": round(ms / max(args.profile_steps, 1), 3)}
        print(json.dumps(line))
    trainer.close()
    if world > 1:
        dist.destroy_process_group()
