"""`bench.py --model trajectoryformer`: one TrajectoryFormer training step (BASELINE configs[4], batch 4 / GPU in the
reference) -- per-frame rotated NMS of eleven frames of detector boxes, IoU linking, motion forecast, hypothesis
generation, point crop + point encoder, box-sequence PointNet, global/local hypothesis encoder, losses, backward,
gradient exchange, AdamW + OneCycle + gradient clipping.  Samples/s over all ranks."""
import json
import os
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(args, rank, local_rank, world, dev):
    import numpy as np

    from ..engine import Trainer
    from .synthetic import synthetic_tracking_batch
    from .trajectoryformer import TrajectoryFormer

    trainer = Trainer(config=os.path.join(ROOT, "configs", "trajectoryformer_waymo_centerpoint.yaml"), device=dev, seed=0,
                      model_cls=TrajectoryFormer, max_iters=6 * (158081 // (4 * 8)))
    np.random.seed(1000 + rank)
    pool = [synthetic_tracking_batch(7000 + 100 * p + rank * args.scenes, args.scenes, device=dev, n_points=args.points,
                                     n_objects=args.objects, n_false=args.objects // 3) for p in range(args.pool)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(max(args.warmup, 1)):
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
    if rank == 0:
        m = trainer.model
        print(json.dumps({
            "metric": "samples/sec TrajectoryFormer Waymo tracking train step", "value": args.scenes * world * args.steps / elapsed,
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "TrajectoryFormer (CenterPoint boxes), %d samples/GPU, %d pts/sample (5 sweeps), %d objects, "
                                   "%d tracks x %d hypotheses x %d points in the last step, hidden 256, 3+3 encoder layers, "
                                   "fwd+bwd+AdamW+OneCycle+clip" % (args.scenes, args.points, args.objects, m.num_track,
                                                                    m.num_hypo_train, m.num_lidar_points),
                       "global_batch": args.scenes * world, "parallelism": "dp%d" % world}}))
    trainer.close()
    if world > 1:
        dist.destroy_process_group()
