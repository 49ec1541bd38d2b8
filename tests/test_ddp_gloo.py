"""CPU, world_size 2 over gloo: the data-parallel path (scenes sharded by rank, DDP gradient
all-reduce, identical parameters after the optimizer step).  The HIP ops are replaced by the oracle
(oracle/cpu_backend.py) because this container has no GPU; on the GPU box the same Trainer runs
over RCCL ("nccl" backend)."""
import os
import socket

import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))   # spawned workers import golden_init too


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, mode):
    os.environ["EFG_DDP_MODE"] = mode
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import oracle  # noqa: F401
    from oracle import cpu_backend

    import copy

    from golden_init import FULL_OVERRIDES, full_inputs

    from efg_amd.engine import Trainer

    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the reduced grid of the full-model golden (128 x 128 x 40 voxels, 16 x 16 BEV tokens): the exchange logic does not
    # depend on the scene size, and a CPU step at the full 188 x 188 token grid costs 10 s apiece
    ov = dict(FULL_OVERRIDES)
    ov.update({"model.transformer.num_queries": 40, "model.transformer.dec_layers": 2})
    tr = Trainer(device="cpu", overrides=ov, seed=0, ddp=True)
    points_list, annos = full_inputs()
    tr.model.noise_generator = torch.Generator().manual_seed(100 + rank)
    # two steps: the flat / bucket buffers are laid out in step 1 and reused in step 2; DDP's static_graph steady
    # state starts at step 2
    with cpu_backend.install():
        for it in range(2):
            scene = (rank + it) % len(points_list)                                 # rank-sharded scenes
            batch = [({"points": torch.from_numpy(points_list[scene])}, {"annotations": copy.deepcopy(annos[scene])})]
            loss_dict, total = tr.step(batch)
    w = tr.model.backbone.extractor.bottom_up.stem.conv1[0].weight
    g = w.grad.detach().clone()
    gathered = [torch.zeros_like(g) for _ in range(world)]
    dist.all_gather(gathered, g)
    flat = torch.cat([p.data.reshape(-1) for p in tr.model.parameters() if p.requires_grad])
    pw = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(pw, flat)
    if rank == 0:
        out["grads_equal"] = bool(torch.equal(gathered[0], gathered[1]))      # all-reduced (averaged) gradient
        out["params_equal"] = bool(torch.equal(pw[0], pw[1]))                  # same update on every rank, every parameter
        out["finite"] = bool(torch.isfinite(total))
        out["grad_norm"] = float(g.norm())
        out["loss"] = float(total)
        out["mode"] = type(tr.grad_sync).__name__.replace("GradientAllReduce", "").lower().replace("bucketed", "bucket") if tr.grad_sync is not None else ("static" if getattr(tr.wrapped, "static_graph", False) else "other")
    dist.destroy_process_group()


def _run(mode):
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(2, port, out, mode), nprocs=2, join=True)
        res = dict(out)
    assert res["finite"] and res["grads_equal"] and res["params_equal"] and res["grad_norm"] > 0
    assert res["mode"] == mode
    return res


@pytest.mark.timeout(900)
def test_two_rank_flat_and_bucketed_exchange_agree(oracle_mod):
    flat, bucket = _run("flat"), _run("bucket")
    assert bucket["grad_norm"] == pytest.approx(flat["grad_norm"], rel=1e-5)  # the same averaged gradient
    assert bucket["loss"] == pytest.approx(flat["loss"], rel=1e-5)


@pytest.mark.timeout(900)
def test_two_rank_torch_ddp_wrapper(oracle_mod):
    _run("static")


def test_exchange_defaults_follow_the_launch_environment(monkeypatch):
    """N > 1 over RCCL: two hardware queues (a collective beside backward doubles the step with eight on this stack) and then
    the overlapped bucket exchange; one rank, or gloo: eight queues and the flat exchange; an explicit setting always wins."""
    from efg_amd import engine

    for env, want in (({"WORLD_SIZE": "8"}, "2"), ({"WORLD_SIZE": "1"}, "8"), ({}, "8"),
                      ({"WORLD_SIZE": "8", "EFG_DIST_BACKEND": "gloo"}, "8"), ({"WORLD_SIZE": "8", "GPU_MAX_HW_QUEUES": "5"}, "5")):
        for k in ("WORLD_SIZE", "EFG_DIST_BACKEND", "GPU_MAX_HW_QUEUES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        engine.configure_hip_runtime()
        assert os.environ["GPU_MAX_HW_QUEUES"] == want, (env, os.environ["GPU_MAX_HW_QUEUES"])
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "2")
    assert engine.default_ddp_mode() == "flat"      # no process group (and gloo would say the same): never the RCCL default


def test_exchange_views_keep_the_parameters_layout():
    """The flat / bucket buffers hand every parameter its reduced gradient as a view in the parameter's OWN strides (the fused
    optimizer refuses a gradient whose layout differs from its parameter's: CenterPoint's channels-last neck / head)."""
    from efg_amd.engine import FlatGradientAllReduce

    port = _free_port()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Conv2d(8, 16, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(16, 4, 1))
        model = model.to(memory_format=torch.channels_last)
        x = torch.randn(2, 8, 12, 12).contiguous(memory_format=torch.channels_last)
        model(x).square().sum().backward()
        want = {n: p.grad.clone() for n, p in model.named_parameters()}
        sync = FlatGradientAllReduce(model, 1)
        sync.reduce()
        for n, p in model.named_parameters():
            assert p.grad.stride() == p.stride(), n
            assert p.grad.data_ptr() >= sync.flat.data_ptr() and torch.equal(p.grad, want[n]), n
        w = model[0].weight
        assert not w.is_contiguous() and w.is_contiguous(memory_format=torch.channels_last)
        torch.optim.AdamW(model.parameters(), lr=1e-3, foreach=True).step()
    finally:
        dist.destroy_process_group()
