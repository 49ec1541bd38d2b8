"""Minimal training engine for the hot path: model + AdamWMulti + (for N > 1) a gradient all-reduce over RCCL.

Counterpart of the slice of efg/engine/trainer.py:168-199,278-305 and efg/engine/hooks.py:68-81 that
surrounds the path: `step()` = zero_grad -> loss_dict = model(batch) -> sum of differentiable losses
-> backward -> one all-reduce of the gradients over xGMI (FlatGradientAllReduce) -> optimizer.step().
One process per GPU; scenes are sharded across ranks, no activation exchange."""
import gc
import os

import numpy as np
import torch
import torch.distributed as dist
from torch.autograd.profiler import record_function

from .config import load_config
from .data.synthetic import make_scene
from .detection3d.optimizer import build_adamw_multi
from .detection3d.voxel_detr import VoxelDETR

DEFAULT_CONFIG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs",
                              "conquer_waymo_res18.yaml")


def init_distributed():
    """torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); backend "nccl" is RCCL on ROCm."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("EFG_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def synthetic_batch(seed_base, scenes, n_points=180000, n_sweeps=1, device=None, n_boxes=40, clutter=0.0):
    """`scenes` (points on `device`, Waymo-style annotations) pairs in the model's input format."""
    batch = []
    for i in range(scenes):
        pts, boxes, labels = make_scene(seed_base + i, n_points=n_points, n_sweeps=n_sweeps, n_boxes=n_boxes,
                                        clutter=clutter)
        pts = torch.from_numpy(pts)
        if device is not None:
            pts = pts.to(device)
        ann = {"gt_boxes": boxes, "labels": labels, "difficulty": np.zeros(len(labels), np.int64),
               "num_points_in_gt": np.full(len(labels), 50, np.int64)}
        sample = {"points": pts}
        if pts.is_cuda:
            # "the points are complete once this event fires": lets the model start voxelization on its geometry
            # stream without ordering it after unrelated work queued on the main stream (voxel_detr.py:_inputs)
            sample["ready_event"] = torch.cuda.Event()
            sample["ready_event"].record(torch.cuda.current_stream(pts.device))
        batch.append((sample, {"annotations": ann}))
    return batch


TUNED_GEMMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gemm_gfx950.csv")


def use_tuned_gemms(path=TUNED_GEMMS):
    """Select the hipBLASLt / rocBLAS solutions recorded by scripts/tune_gemm.py for the step's dense fp32 GEMM
    shapes (PyTorch TunableOp, tuning itself OFF: unknown shapes keep the library default).  Pure algorithm
    selection -- same fp32 arithmetic; entries are validated against the ROCm / hipBLASLt build and ignored on
    mismatch.  EFG_TUNED_GEMMS=0 disables it."""
    path = os.environ.get("EFG_TUNED_GEMMS_FILE", path)
    if os.environ.get("EFG_TUNED_GEMMS", "1") == "0" or not os.path.exists(path):
        return False
    import torch.cuda.tunable as tunable

    tunable.enable(True)
    tunable.tuning_enable(False)
    tunable.set_filename(path)
    return True


def configure_hip_runtime():
    """Process-level HIP runtime settings of a TRAINING process (bench.py and Trainer call this; importing efg_amd
    does not touch the environment).

    GPU_MAX_HW_QUEUES: the HIP runtime multiplexes all streams of a process onto this many hardware queues.
      * ONE rank (no collective beside backward): 8.  The step uses the main stream, the high-priority geometry stream
        (voxelization and sparse-conv site counts, whose read-backs the host waits on) and the shared side stream; with the
        runtime's default of 4 and a live communicator the geometry stream shared a queue with unrelated work: +1.4 ms/step
        (scripts/ubench/ddp_modes.py none vs none:comm: 35.4 -> 36.8 ms; 35.8 / 35.8 with 8 queues).
      * MORE than one rank over RCCL: 2.  A collective that runs BESIDE backward -- the bucketed, overlapped exchange that
        `north_star` and the reference's DDP reducer ask for -- doubles the step with 8 queues on this stack (bucket 61-66 ms,
        torch DDP static_graph 72 ms against 31.3 flat; measured in rounds 2, 4, 5 and 6), and does NOT with 2 or 3: flat
        31.3-31.4, bucket 31.5-31.7 (profiles/r05d_ddp_modes_hw_queues.txt), and there it hides wire time --
        scripts/ubench/ddp_overlap_probe.py adds a spin kernel of the all-reduce's length behind every collective: 3 ms of
        simulated wire per step cost the flat exchange +2.9 ms and the bucketed one +1.4 (profiles/r06_ddp_overlap_probe.txt).
        The flat exchange itself loses nothing at 2 queues.
    The runtime reads the variable when it initialises, so this only has an effect before the first HIP call of the
    process; an explicit setting in the environment wins.  Returns True if the setting can still take effect."""
    try:
        world = int(os.environ.get("WORLD_SIZE", "1") or 1)
    except ValueError:
        world = 1
    rccl = os.environ.get("EFG_DIST_BACKEND", "nccl") == "nccl"
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "2" if world > 1 and rccl else "8")
    return not torch.cuda.is_initialized()


def default_ddp_mode():
    """The gradient exchange a Trainer uses when neither its argument nor EFG_DDP_MODE names one: `bucket` (three all-reduces
    issued DURING backward, overlapped with it) over RCCL when the process runs with the few hardware queues that make a
    collective beside backward safe (configure_hip_runtime sets 2 for WORLD_SIZE > 1), else `flat` (one all-reduce after
    backward: gloo, or a process whose environment pinned more queues)."""
    try:
        queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        queues = 4
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl" and queues <= 3:
        return "bucket"
    return "flat"


def limit_host_threads():
    """The host side of a step is ~2000 kernel launches from two Python threads; nothing in it is a parallel CPU
    loop worth more than a few cores.  Left alone, OpenMP sizes its pool to the machine (256 hardware threads on
    the GPU box) and the spin-waiting workers of any stray parallel region burn the container's CPU quota
    (16 cores per 100 ms period here): the cgroup then throttles the WHOLE process for tens of ms -- 13-21 throttled
    periods per 45-step run, sporadic 25-45 ms stalls of the launch threads, 37.6-41.2 ms/step instead of 37.1-37.3
    (scripts/ubench/throttle_ab.sh).  torchrun already exports OMP_NUM_THREADS=1 for N > 1; this does the same job
    for a bare `python bench.py`.  EFG_HOST_THREADS / OMP_NUM_THREADS override."""
    if "OMP_NUM_THREADS" in os.environ and "EFG_HOST_THREADS" not in os.environ:
        return int(os.environ["OMP_NUM_THREADS"])
    n = int(os.environ.get("EFG_HOST_THREADS", "4"))
    torch.set_num_threads(n)
    return n


def _weights_written_behind_autograd():
    """Weights were just written through `.data` (the parameter broadcast of an exchange object or of torch DDP's
    constructor, a checkpoint load): neither `Parameter._version` nor the optimizer's post-step hook has moved, so the
    packed MFMA copies the sparse convolutions cache per parameter must be dropped by hand (spconv.weights_updated) --
    otherwise forward / dgrad keep multiplying the copy of the OLD weights while wgrad reads the live ones."""
    from . import spconv

    spconv.weights_updated()


class FlatGradientAllReduce:
    """The data-parallel exchange step: ONE all-reduce (mean) of all gradients after backward.

    Replaces the reference's DistributedDataParallel wrapper (efg/engine/trainer.py:191-198).  DDP hangs a hook on
    each of the ~300 parameters, copies every gradient into a bucket (one small kernel per parameter) and launches
    an all-reduce per bucket so that communication overlaps with backward.  On this step that machinery costs more
    than it can hide: 69 MB of gradients are ~0.4 ms on the xGMI ring, while the wrapper adds 2-3.5 ms per step
    (scripts/ubench/ddp_modes.py, 1 rank on the RCCL backend: bare model 35.4-35.8 ms, `static_graph` DDP 38.9 ms,
    this 35.7-36.3 ms) -- the step is close to host-bound and the hooks sit on the backward thread's critical path.
    Here the gradients are packed by one multi-tensor copy into a persistent flat buffer, reduced by one RCCL call
    (a single large collective: what the point-to-point xGMI links like), and the optimizer reads them through views
    of that buffer (no copy back): 0.23 ms of device time and ~1.1 ms of host time per step
    (scripts/ubench/flat_reduce_cost.py).

    Which parameters are exchanged: those that received a gradient in the first step (the unused FPN levels never
    do, and exchanging 1.3 M zeros for them is wasted wire).  The set is fixed by the graph, not by the data; it is
    nevertheless CHECKED across ranks on that first step (a mismatch would desynchronise the flat buffers: the
    reference runs DDP with find_unused_parameters=True for this reason, $CQ/config.yaml:182-183), and a parameter of
    the set that has no gradient in a later step contributes zeros instead of crashing the pack."""

    def __init__(self, model, world):
        self.model, self.world = model, world
        self.params = self.flat = self.views = None
        self.found_inf = None   # in: this rank's flag (set by Trainer.step before backward); out: any rank's
        self.avg = dist.get_backend() == "nccl"  # gloo has no AVG
        # every rank starts from rank 0's parameters and buffers (DDP's constructor does the same)
        tensors = [p.data for p in model.parameters()] + [b.data for b in model.buffers()]
        with torch.no_grad():
            try:
                dist._broadcast_coalesced(dist.group.WORLD, tensors, 256 * 1024 * 1024, 0)
            except (AttributeError, RuntimeError):
                for t in tensors:
                    dist.broadcast(t, 0)
        _weights_written_behind_autograd()

    @torch.no_grad()
    def reduce(self):
        if self.params is None:
            named = [(n, p) for n, p in self.model.named_parameters() if p.requires_grad]
            used = torch.tensor([1.0 if p.grad is not None else 0.0 for _, p in named], device=named[0][1].device)
            lo, hi = used.clone(), used.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if not torch.equal(lo, hi):
                bad = [named[i][0] for i in torch.nonzero(lo != hi).flatten().tolist()]
                raise RuntimeError("FlatGradientAllReduce: ranks disagree on which parameters receive a gradient "
                                   "(%d parameters, e.g. %s); use EFG_DDP_MODE=find_unused" % (len(bad), bad[:3]))
            self.params = [p for _, p in named if p.grad is not None]
            total = sum(p.numel() for p in self.params)
            # + 1: the step's `found_inf` flag travels with the gradients (see Trainer.step)
            self.flat = torch.zeros(total + 1, dtype=self.params[0].dtype, device=self.params[0].device)
            self.views, off = [], 0
            for p in self.params:
                self.views.append(_flat_view(self.flat, off, p))
                off += p.numel()
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):  # a data-dependent branch skipped a parameter this step: it sends zeros
            grads = [v.zero_() if g is None else g for g, v in zip(grads, self.views)]
        torch._foreach_copy_(self.views, grads)  # multi-tensor pack into the flat buffer
        _pack_found_inf(self.flat, self.found_inf)
        if self.avg:
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(self.flat)
            self.flat.div_(self.world)
        for p, v in zip(self.params, self.views):
            p.grad = v
        self.found_inf = _unpack_found_inf(self.flat, self.found_inf)


def _flat_view(buf, off, p):
    """The slice of an exchange buffer that stands for `p`'s gradient, in `p`'s OWN memory layout: the view becomes `p.grad`,
    and the fused optimizer kernel walks parameter, gradient and moments as flat memory (it refuses lists whose strides
    differ) -- a channels-last convolution weight (CenterPoint's neck and head on the GPU) needs a channels-last view."""
    n = p.numel()
    if p.dim() > 1:
        expect = 1     # dense and non-overlapping <=> the strides are a permutation of a contiguous tensor's
        for size, stride in sorted(zip(p.shape, p.stride()), key=lambda t: (t[1], t[0])):
            if size != 1 and stride != expect:
                break
            expect *= size
        else:
            return buf[off:off + n].as_strided(p.shape, p.stride())
    return buf[off:off + n].view_as(p)


def _pack_found_inf(buf, flag):
    """Last element of an exchange buffer := this rank's "the loss is not finite" flag (0 / 1)."""
    if flag is None:
        buf[-1:].zero_()
    else:
        buf[-1:].copy_(flag.reshape(1))


def _unpack_found_inf(buf, flag):
    """After the SUM / AVG all-reduce the element is > 0 on EVERY rank iff ANY rank raised it: all ranks skip the
    update together (a rank-local flag lets the failing rank skip while the others write the NaN gradients it sent
    into their weights -- the replicas diverge).  Stays on the device."""
    if flag is None:
        return None
    return (buf[-1:] > 0).to(torch.float32)


def build_optimizer(cfg, model):
    """`solver.optimizer.type`: AdamWMulti (ConQueR / Voxel-DETR: per-group rates, $CQ/modules/optimizer.py) or AdamW
    (CenterPoint, efg/solver/optimizers.py:23-40); torch's fused multi-tensor update on the GPU."""
    oc = dict(cfg.solver.optimizer)
    kind = oc.pop("type", "AdamWMulti")
    if kind == "AdamWMulti":
        return build_adamw_multi(cfg, model)
    if kind != "AdamW":
        raise ValueError("optimizer %r is not mirrored (AdamWMulti | AdamW)" % kind)
    oc["betas"] = tuple(oc["betas"])
    params = [p for p in model.parameters() if p.requires_grad]
    if all(p.is_cuda for p in params):
        from .detection3d.optimizer import CachedFusedAdamW

        return CachedFusedAdamW([{"params": params}], **oc)
    return torch.optim.AdamW(params, **oc)


def _scheduler_block(cfg):
    """`solver.lr_scheduler`, or None when the config has no scheduler.  The merged defaults always carry a block of
    placeholders ({max_epochs: None, max_iters: None}): a block that names no type and no length IS "no scheduler"."""
    sc = cfg.solver.get("lr_scheduler") if hasattr(cfg.solver, "get") else None
    if not sc:
        return None
    if not sc.get("type") and not sc.get("max_iters") and not sc.get("max_epochs"):
        return None
    return sc


def resolve_max_iters(cfg, max_iters=None, iters_per_epoch=None):
    """Length of the schedule, per CONFIG, as the reference's trainer derives it (efg/engine/trainer.py:158-161:
    `max_iters = len(dataloader) * lr_scheduler.max_epochs`, written back as lr_scheduler.max_iters / epoch_iters).
    Precedence: the `max_iters` argument; `solver.lr_scheduler.max_iters`; `max_epochs` x (`iters_per_epoch` argument
    = len(loader), else `solver.lr_scheduler.epoch_iters` from the YAML).  No silent fallback: a config that gives
    neither is an error (a ConQueR-sized constant would give CenterPoint's 36-epoch run the wrong rate and momentum)."""
    if max_iters is not None:
        return int(max_iters)
    sc = _scheduler_block(cfg)
    if not sc:
        return None
    if sc.get("max_iters"):
        return int(sc["max_iters"])
    per_epoch = iters_per_epoch if iters_per_epoch is not None else sc.get("epoch_iters")
    if sc.get("max_epochs") and per_epoch:
        return int(sc["max_epochs"]) * int(per_epoch)
    raise ValueError("solver.lr_scheduler needs max_iters, or max_epochs together with epoch_iters (or pass max_iters= / "
                     "iters_per_epoch=len(loader) to Trainer): the schedule length cannot be guessed")


def build_one_cycle(cfg, optimizer, max_iters):
    """`solver.lr_scheduler: {type: OneCycle, ...}` -> torch OneCycleLR exactly as efg/solver/lr_schedulers.py:222-237
    builds it: max_lr = solver.optimizer.lr for EVERY parameter group (the scalar overrides the per-group rates of
    AdamWMulti -- kept, it is what the reference trains with), total_steps = max_iters, cycled Adam beta1 between
    base_momentum and max_momentum.  None when the config has no scheduler."""
    sc = _scheduler_block(cfg)
    if not sc:
        return None
    if sc.get("type", "OneCycle") != "OneCycle":
        raise ValueError("only the OneCycle scheduler of the ConQueR / Voxel-DETR configs is mirrored, got %r" % sc.type)
    kw = {k: sc[k] for k in ("pct_start", "base_momentum", "max_momentum", "div_factor", "final_div_factor",
                             "anneal_strategy", "three_phase") if k in sc}
    return torch.optim.lr_scheduler.OneCycleLR(optimizer, cfg.solver.optimizer.lr, total_steps=int(max_iters), **kw)


class BucketedGradientAllReduce:
    """EFG_DDP_MODE=bucket: the gradient exchange in THREE flat buckets, each reduced as soon as backward has produced
    it, on a communication stream that depends on the compute stream -- the overlap the reference gets from DDP's
    reducer (efg/engine/trainer.py:191-198) without a hook per parameter (~300 Python callbacks and bucket copies on
    the backward thread of a step that is host-bound already).

    Buckets follow the order backward finishes them: [transformer + heads + contrastive MLPs] -> [input projection +
    FPN] -> [sparse backbone].  The trigger for the first two is ONE tensor hook each, on the activation that enters
    that part of the model (the projected BEV tokens; the dense BEV maps of the backbone): when its gradient exists,
    every backward node of the part has run and so have their AccumulateGrad nodes (the engine gives those the highest
    priority).  Membership is fixed on the first step (parameters that received a gradient, validated across ranks;
    that step reduces all three buckets after backward); a member whose gradient is missing later sends zeros, so
    every rank always issues the same three collectives with the same sizes.
    `reduce()` after backward handles the last bucket and joins the communication stream."""

    def __init__(self, model, world):
        self.model, self.world = model, world
        self.flat = FlatGradientAllReduce(model, world)  # (broadcasts the initial parameters)
        self.avg = self.flat.avg
        self.groups = None
        self.layout = None  # {bucket: (params, flat buffer, views)}, frozen on the first step
        self.comm = None
        self.pending = []
        self.done = set()
        self.found_inf = None   # as in FlatGradientAllReduce; rides in the LAST bucket (backbone)

    def _build(self):
        names = {"transformer": [], "neck": [], "backbone": []}
        for n, p in self.model.named_parameters():
            if not p.requires_grad:
                continue
            if n.startswith("backbone.extractor.bottom_up."):
                names["backbone"].append(p)
            elif n.startswith("backbone.") or n.startswith("input_proj."):
                names["neck"].append(p)
            else:
                names["transformer"].append(p)
        self.groups = names

    def _launch(self, key):
        """Pack bucket `key` and all-reduce it on the communication stream.  The bucket's membership and layout are FIXED
        (see _freeze): every rank issues the same collective with the same element count every step; a member without
        a gradient at this moment contributes zeros."""
        if key in self.done:
            return
        self.done.add(key)
        if self.layout is None:   # first step: nothing is launched from the hooks, reduce() freezes the membership
            return
        params, buf, views = self.layout[key]
        if not params and key != "backbone":
            return
        dev = buf.device
        main = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        grads = [p.grad for p in params]
        missing = [i for i, g in enumerate(grads) if g is None]
        if missing:
            grads = [v.zero_() if g is None else g for g, v in zip(grads, views)]
        torch._foreach_copy_(views, grads)  # on the compute stream, right behind the producers
        if key == "backbone":
            _pack_found_inf(buf, self.found_inf)
        if main is not None:
            if self.comm is None:
                from .streams import side_stream

                self.comm = side_stream(dev, "exchange")   # the shared side stream (streams.py: why not one of its own)
            self.comm.wait_stream(main)
            with torch.cuda.stream(self.comm):
                work = dist.all_reduce(buf, op=dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM, async_op=True)
        else:
            work = dist.all_reduce(buf, op=dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM, async_op=True)
        self.pending.append((work, buf, params, views, missing))

    def _freeze(self):
        """First step, after backward: the members of each bucket are the parameters that received a gradient, checked
        across ranks with a MIN / MAX all-reduce exactly as FlatGradientAllReduce does (ranks that disagreed would issue
        collectives of different sizes: a hang, or silently mis-paired gradients).  Buffers and views are built once."""
        named = [(key, p) for key in ("transformer", "neck", "backbone") for p in self.groups[key]]
        dev = named[0][1].device
        used = torch.tensor([1.0 if p.grad is not None else 0.0 for _, p in named], device=dev)
        lo, hi = used.clone(), used.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise RuntimeError("BucketedGradientAllReduce: ranks disagree on which parameters receive a gradient "
                               "(%d parameters); use EFG_DDP_MODE=find_unused" % int((lo != hi).sum()))
        self.layout = {}
        for key in ("transformer", "neck", "backbone"):
            params = [p for p in self.groups[key] if p.grad is not None]
            total = sum(p.numel() for p in params) + (1 if key == "backbone" else 0)   # + the found_inf element
            buf = torch.zeros(total, dtype=params[0].dtype if params else torch.float32, device=dev)
            views, off = [], 0
            for p in params:
                views.append(_flat_view(buf, off, p))
                off += p.numel()
            self.layout[key] = (params, buf, views)

    def watch(self, key, tensor):
        """Called by the model hooks installed in Trainer: `tensor` is the activation entering part `key`."""
        if tensor.requires_grad:
            tensor.register_hook(lambda g, key=key: (self._launch(key), None)[1])

    @torch.no_grad()
    def begin_step(self):
        if self.groups is None:
            self._build()
        self.done.clear()
        self.pending.clear()

    @torch.no_grad()
    def reduce(self):
        if self.layout is None:
            self._freeze()
            self.done.clear()
        for key in ("transformer", "neck", "backbone"):  # whatever the hooks did not launch (always: backbone)
            self._launch(key)
        for work, buf, params, views, missing in self.pending:
            # a gradient that was absent when its bucket was packed and exists now arrived AFTER the bucket's hook fired:
            # the hook placement is wrong for this model -- fail loudly, the exchange would silently drop it
            late = [i for i in missing if params[i].grad is not None]
            if late:
                raise RuntimeError("BucketedGradientAllReduce: %d gradients were produced after their bucket had been "
                                   "packed (the bucket hook fired too early); use EFG_DDP_MODE=flat" % len(late))
            work.wait()  # orders the compute stream after the collective (device side)
            if not self.avg:
                buf.div_(self.world)
            for p, v in zip(params, views):
                p.grad = v
        if self.comm is not None:
            torch.cuda.current_stream().wait_stream(self.comm)
        self.found_inf = _unpack_found_inf(self.layout["backbone"][1], self.found_inf)


class Trainer:
    """step() = the reference's `DefaultTrainer.step` + `Optimization.after_step` + `LRScheduler.after_step`
    (efg/engine/trainer.py:278-305, efg/engine/hooks.py:68-81,118-121): zero_grad, forward, sum of the differentiable
    losses, non-finite check, backward, gradient exchange, optional clipping, optimizer step, scheduler step."""

    def __init__(self, config=None, overrides=None, device=None, seed=0, ddp=None, max_iters=None, model_cls=None,
                 iters_per_epoch=None, ddp_mode=None):
        cfg = load_config(config or DEFAULT_CONFIG, overrides)
        if str(cfg.model.device if device is None else device).startswith("cuda"):
            configure_hip_runtime()
            limit_host_threads()
        if device is not None:
            cfg.model.device = str(device)
        if str(cfg.model.device).startswith("cuda"):
            use_tuned_gemms()
            # let MIOpen time its solvers for the one dense 3x3 BEV convolution (FPN output, 256 -> 256 at 188^2)
            # instead of taking the immediate-mode pick: fwd 1.24 -> 0.66 ms, bwd-data 0.85 -> 0.66 ms per step
            if os.environ.get("EFG_MIOPEN_FIND", "1") == "1":
                torch.backends.cudnn.benchmark = True
        torch.manual_seed(seed)
        self.cfg = cfg
        from .streams import create_side_streams

        create_side_streams(torch.device(str(cfg.model.device)))   # before anything else asks for a pool stream (streams.py)
        self.model = (model_cls or VoxelDETR)(cfg)
        self.model.train()
        self.optimizer = build_optimizer(cfg, self.model)
        self.max_iters = resolve_max_iters(cfg, max_iters, iters_per_epoch)
        self.lr_scheduler = build_one_cycle(cfg, self.optimizer, self.max_iters)
        gc_cfg = cfg.solver.get("grad_clipper") if hasattr(cfg.solver, "get") else None
        self.grad_clipper = gc_cfg if (gc_cfg and gc_cfg.get("enabled")) else None
        self.anomaly_every = int(os.environ.get("EFG_ANOMALY_EVERY", "50"))
        self._nonfinite = None
        self._gc_state = None
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        use_ddp = (world > 1) if ddp is None else ddp
        self.wrapped = self.model
        self._steps = 0
        self._manual_gc = (self.model.device.type == "cuda" and os.environ.get("EFG_MANUAL_GC", "1") != "0")
        self.grad_sync = None
        self.ddp_mode = None
        if use_ddp:
            # "bucket" (the default over RCCL, default_ddp_mode): three flat buckets, each all-reduced on a communication
            # stream as soon as backward has produced it -- the exchange OVERLAPS backward, as the reference's DDP reducer
            # does (efg/engine/trainer.py:191-198) and BASELINE.json's north_star asks (BucketedGradientAllReduce).  It
            # needs the process to run with 2-3 hardware queues (configure_hip_runtime does that for WORLD_SIZE > 1): with
            # 8, a collective beside backward doubles the step on this stack.
            # "flat" (gloo, or more hardware queues pinned by the environment): one all-reduce of a flat gradient buffer
            # after backward (FlatGradientAllReduce).
            # "static" / "find_unused" / "plain": torch DistributedDataParallel as in the reference, with
            # static_graph=True / find_unused_parameters=True ($CQ/config.yaml:183) / neither.  The literal
            # find_unused_parameters setting costs +14 ms/step: with locally unused parameters (the skipped FPN
            # levels) DDP makes a BLOCKING D2H copy of its "used" bitmap at the end of every backward.
            mode = ddp_mode or os.environ.get("EFG_DDP_MODE") or default_ddp_mode()
            self.ddp_mode = mode
            if mode == "bucket" and not hasattr(self.model, "grad_watch"):
                mode = self.ddp_mode = "flat"   # a model without the bucket hooks (CenterPoint, TrajectoryFormer)
            if mode == "flat":
                self.grad_sync = FlatGradientAllReduce(self.model, world)
            elif mode == "bucket":
                self.grad_sync = BucketedGradientAllReduce(self.model, world)
                self.model.grad_watch = self.grad_sync.watch
            else:
                kw = {}
                if mode == "find_unused":
                    kw["find_unused_parameters"] = True
                elif mode == "static":
                    kw["static_graph"] = True
                elif mode != "plain":
                    raise ValueError("EFG_DDP_MODE must be flat, bucket, static, find_unused or plain, got %r" % mode)
                dev_ids = [torch.cuda.current_device()] if self.model.device.type == "cuda" else None
                # torch DDP looks for the tensors of the forward output with pytree, which knows `dict` but treats a dict
                # SUBCLASS as an opaque leaf: with the model's LossDict it would see no output at all (static_graph: the
                # delayed all-reduce is never triggered; find_unused: every parameter counts as unused).  These modes get
                # the plain dict of scalar terms -- and the trainer sums its values, as the reference does.
                self.model.plain_loss_dict = True
                self.wrapped = torch.nn.parallel.DistributedDataParallel(
                    self.model, device_ids=dev_ids, broadcast_buffers=False,
                    bucket_cap_mb=50, gradient_as_bucket_view=True, **kw)
                _weights_written_behind_autograd()   # (its constructor broadcasts rank 0's parameters through .data)

    def _collect_garbage(self):
        """Python's cyclic collector runs a few hundred times per step on the containers autograd creates and
        costs ~3 ms of host time per step (35.5 -> 32 ms, scripts/ubench/jitter.py); the step frees its graph by
        reference counting.  At the first step (always a warm-up step): freeze what exists, switch the automatic
        collector off; afterwards collect by hand every 100 steps.  EFG_MANUAL_GC=0 leaves the interpreter alone."""
        if self._steps == 1:
            self._gc_state = gc.isenabled()
            gc.collect()
            gc.freeze()
            gc.disable()
        elif self._steps % 100 == 0:
            gc.collect()

    def close(self):
        """Give the interpreter its cyclic collector back (the process may go on to do other things) and run the
        pending anomaly check."""
        if self._gc_state is not None:
            gc.unfreeze()
            if self._gc_state:
                gc.enable()
            self._gc_state = None
        self._check_anomaly(force=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _check_anomaly(self, losses=None, force=False):
        """The reference raises FloatingPointError when the summed loss is not finite (trainer.py:307-311) and scipy
        raises on an infeasible cost matrix (matcher.py:89) -- both by reading a device value on the host EVERY step.
        Here both conditions are accumulated on the device and read every `anomaly_every` steps (EFG_ANOMALY_EVERY,
        default 50; the first step always): same failure, at most that many steps late, no per-step drain of the
        stream.  Unmatched queries never index out of range in between (csrc/det_loss.hip skips them)."""
        if losses is not None:  # running sum: one tiny kernel per step; a NaN / Inf stays non-finite in the sum
            self._nonfinite = losses.detach() if self._nonfinite is None else self._nonfinite + losses.detach()
        if not (force or self._steps == 1 or (self.anomaly_every > 0 and self._steps % self.anomaly_every == 0)):
            return
        lsap_bad = None
        if self.model.device.type == "cuda":
            from .operators.assignment import take_failures

            lsap_bad = take_failures(self.model.device)
            import ctypes

            from . import _lib

            ring = ctypes.c_int64(0)
            with torch.cuda.device(self.model.device):   # the ring (and its error word) of THIS model's device
                _lib.check(_lib.lib().efg_ticket_ring_errors(ctypes.byref(ring), 1))
            if ring.value:
                raise RuntimeError("efg_amd: %d ticket counters of the fused column / focal sums were found not at rest at or "
                                   "before iteration=%d: bias gradients or loss sums since then are unreliable" % (
                                       ring.value, self._steps))
        flag, self._nonfinite = self._nonfinite, None
        if flag is not None and not bool(torch.isfinite(flag)):
            raise FloatingPointError("Loss became infinite or NaN at or before iteration=%d!" % self._steps)
        if lsap_bad is not None and int(lsap_bad) > 0:
            raise FloatingPointError("Hungarian matching met %d infeasible cost matrices (non-finite costs) at or "
                                     "before iteration=%d" % (int(lsap_bad), self._steps))

    def step(self, batch):
        self._steps += 1
        if self._manual_gc:
            self._collect_garbage()
        # == optimizer.zero_grad(set_to_none=True) over the optimizer's own parameters, without its per-call bookkeeping
        # (0.4 ms of host time per step for 261 parameters)
        zg = self.__dict__.get("_zero_grad_params")
        if zg is None or zg[0] != sum(len(g["params"]) for g in self.optimizer.param_groups):
            plist = [p for g in self.optimizer.param_groups for p in g["params"]]
            zg = self.__dict__["_zero_grad_params"] = (len(plist), plist)
        for p in zg[1]:
            if p.grad is not None:
                p.grad = None
        if hasattr(self.grad_sync, "begin_step"):
            self.grad_sync.begin_step()
        with record_function("efg::forward"):
            loss_dict = self.wrapped(batch)
            if hasattr(loss_dict, "total"):
                # detection3d.losses.LossDict: the terms are views of a few vectors; summing the vectors keeps the
                # ~32 select nodes of the scalar entries (a zero-fill + copy each in backward) out of the graph
                losses = loss_dict.total()
            else:
                # one stack + sum instead of 31 chained scalar adds (and as many backward nodes)
                losses = torch.stack([v for v in loss_dict.values() if torch.is_tensor(v) and v.requires_grad]).sum()
        if losses.device.type == "cpu":
            if not bool(torch.isfinite(losses)):   # the reference's check (trainer.py:307-311); free on the host
                raise FloatingPointError("Loss became infinite or NaN at iteration=%d!" % self._steps)
        elif hasattr(self.optimizer, "_plan"):
            # on the GPU the error is REPORTED every `anomaly_every` steps (_check_anomaly), but a non-finite step never
            # reaches the weights: the fused AdamW skips its update on a device-side flag
            self.optimizer.found_inf = torch.logical_not(torch.isfinite(losses.detach())).to(torch.float32).reshape(1)
            if hasattr(self.grad_sync, "found_inf"):   # exchanged with the gradients: every rank skips, or none
                self.grad_sync.found_inf = self.optimizer.found_inf
        with record_function("efg::backward"):
            losses.backward()
            if self.grad_sync is not None:
                self.grad_sync.reduce()
                if getattr(self.grad_sync, "found_inf", None) is not None and hasattr(self.optimizer, "_plan"):
                    self.optimizer.found_inf = self.grad_sync.found_inf
            elif (self.wrapped is not self.model and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
                  and getattr(self.optimizer, "found_inf", None) is not None and hasattr(self.optimizer, "_plan")):
                # torch-DDP modes (static / find_unused / plain): the gradients were averaged inside backward, the flag
                # was not -- one 1-element MAX so that every rank skips the update or none does (a rank that skipped alone
                # would leave its replica behind the others for good)
                dist.all_reduce(self.optimizer.found_inf, op=dist.ReduceOp.MAX)
        with record_function("efg::optimizer"):
            if self.grad_clipper is not None:  # hooks.py:74-79 (disabled in the ConQueR / Voxel-DETR configs)
                params = [p for p in self.model.parameters() if p.grad is not None]
                if self.grad_clipper.clip_type == "norm":
                    torch.nn.utils.clip_grad_norm_(params, **dict(self.grad_clipper.params))
                elif self.grad_clipper.clip_type == "value":
                    torch.nn.utils.clip_grad_value_(params, **dict(self.grad_clipper.params))
            self.optimizer.step()
            if self.lr_scheduler is not None:
                self.lr_scheduler.step()
        self._check_anomaly(losses)
        return loss_dict, losses
