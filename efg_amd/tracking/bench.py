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

    # The parameter-free half of the step (NMS, linking, forecast, hypotheses, point crop, targets) runs as the
    # DeviceLoader's collate, two batches ahead on the loader's thread and stream; the loader is long enough to keep
    # preparing through the whole timed region (it still prepares batches K+1, K+2 while step K trains), so the K timed
    # steps contain K preparations.  EFG_TF_LOADER=0: preparation inside the step, on the model's side stream.
    warmup = max(args.warmup, 1)
    loader = None
    if os.environ.get("EFG_TF_LOADER", "1") != "0":
        from ..data.loader import DeviceLoader

        def produce(i):
            sample, info = pool[(i // args.scenes) % len(pool)][i % args.scenes]
            return [dict(sample[0])], info    # a fresh sample dict: the loader attaches per-batch state to it

        loader = DeviceLoader(produce, batch_size=args.scenes, length=warmup + args.steps + 2, device=dev, depth=2,
                              collate=trainer.model.prepare)
        batches = iter(loader)
    else:
        batches = (pool[i % len(pool)] for i in range(warmup + args.steps))
    for w in range(warmup):
        trainer.step(next(batches))
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        trainer.step(next(batches))
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
                       "preparation": "DeviceLoader, 2 batches ahead" if loader is not None else "in step, side stream",
                       "global_batch": args.scenes * world, "parallelism": "dp%d" % world}}))
    if loader is not None:
        loader.close()
    trainer.close()
    if world > 1:
        dist.destroy_process_group()
