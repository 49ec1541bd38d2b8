"""Worker of tests/test_multirank_gpu.py (launched by torch.distributed.run, 2 ranks on ONE GPU over gloo -- RCCL
refuses two ranks on the same device): three data-parallel training steps on rank-sharded scenes through the real HIP
path, then every rank's gradients and parameters are compared."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efg_amd.engine import Trainer, init_distributed, synthetic_batch  # noqa: E402

rank, local_rank, world = init_distributed()
dev = torch.device("cuda", torch.cuda.current_device())   # gloo: both ranks on cuda:0; RCCL: one device per rank
ov = {"model.transformer.num_queries": 60, "model.transformer.enc_layers": 1, "model.transformer.dec_layers": 2}
tr = Trainer(device=dev, overrides=ov, seed=0, ddp=True, max_iters=50)   # exchange: EFG_DDP_MODE (default: flat over gloo, bucket over RCCL)
tr.model.noise_generator = torch.Generator().manual_seed(100 + rank)
for it in range(3):
    batch = synthetic_batch(700 + 10 * it + rank, 1, n_points=30000, n_boxes=12, device=dev)  # different scenes per rank
    loss_dict, total = tr.step(batch)
torch.cuda.synchronize()
names = ["backbone.extractor.bottom_up.stem.conv1.0.weight", "backbone.extractor.fpn_lateral3.weight",
         "transformer.decoder.layers.1.linear1.weight"]
params = dict(tr.model.named_parameters())
g = torch.cat([params[n].grad.reshape(-1) for n in names])
gs = [torch.zeros_like(g) for _ in range(world)]
dist.all_gather(gs, g)
flat = torch.cat([p.data.reshape(-1) for p in tr.model.parameters() if p.requires_grad])
ps = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(ps, flat)
# A step whose loss is not finite on ONE rank only: the flag travels with the gradients, EVERY rank skips the update
# (rank-local handling would let rank 0 apply the NaN gradients rank 1 sent: ADVICE r3, engine.py found_inf).
before = flat.clone()
inner = tr.wrapped
if rank == 1:
    tr.wrapped = lambda b: {k: (v * float("nan") if torch.is_tensor(v) and v.requires_grad else v) for k, v in inner(b).items()}
tr.step(synthetic_batch(900 + rank, 1, n_points=30000, n_boxes=12, device=dev))
tr.wrapped = inner
torch.cuda.synchronize()
after = torch.cat([p.data.reshape(-1) for p in tr.model.parameters() if p.requires_grad])
skipped = torch.tensor([1.0 if torch.equal(before, after) else 0.0], device=dev)
dist.all_reduce(skipped, op=dist.ReduceOp.MIN)
tr._nonfinite = None          # (the deferred anomaly report of rank 1 is not part of this check)
if rank == 0:
    ok = torch.equal(gs[0], gs[1]) and torch.equal(ps[0], ps[1]) and bool(torch.isfinite(total)) and float(g.norm()) > 0
    print("DDP_GPU_%s mode=%s grad_norm=%.6e loss=%.5f nan_step_skipped_on_all_ranks=%d" % (
        "OK" if ok else "MISMATCH", type(tr.grad_sync).__name__, float(g.norm()), float(total), int(skipped.item())))
tr.close()
dist.destroy_process_group()
