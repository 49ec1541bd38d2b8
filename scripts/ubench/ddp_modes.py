"""How much does the DDP wrapper itself cost per step (1 rank, RCCL backend, so no wire time)?

    python scripts/ubench/ddp_modes.py [mode ...]     modes: none flat bucket find_unused static plain; `:comm` creates the communicator without using it

`find_unused` is the reference's setting (find_unused_parameters: True); DDP then all-reduces a "used" bitmap and,
because the skipped FPN levels leave locally unused parameters, makes a BLOCKING D2H copy of it at the end of
every backward.  `static` = static_graph=True (the Trainer's default)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efg_amd.engine import configure_hip_runtime  # noqa: E402

configure_hip_runtime()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from efg_amd.engine import Trainer, synthetic_batch  # noqa: E402


def run(mode, steps=20, warmup=6):
    mode, _, opts = mode.partition(":")  # e.g. none:comm
    os.environ["EFG_DDP_MODE"] = mode
    if "comm" in opts:  # create the RCCL communicator without using it in the step
        dist.all_reduce(torch.zeros(4, device="cuda:0"))
    tr = Trainer(device="cuda:0", overrides={"model.transformer.num_queries": 1000}, seed=0, ddp=(mode != "none"))
    pool = [synthetic_batch(2000 + 100 * p, 2, device="cuda:0") for p in range(2)]
    for w in range(warmup):
        tr.step(pool[w % 2])
    torch.cuda.synchronize()

    def cpu_usec():
        try:
            with open("/sys/fs/cgroup/cpu.stat") as f:
                return int(f.readline().split()[1])
        except OSError:
            return 0

    c0 = cpu_usec()
    t0 = time.perf_counter()
    for s in range(steps):
        tr.step(pool[s % 2])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("    container CPU: %.2f cores busy during the timed steps" % ((cpu_usec() - c0) / 1e6 / dt))
    return 1000 * dt / steps


if __name__ == "__main__":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    for mode in (sys.argv[1:] or ["none", "flat", "static", "find_unused", "none"]):
        print("%-12s %.2f ms/step" % (mode, run(mode)), flush=True)
    dist.destroy_process_group()
