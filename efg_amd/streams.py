"""Process-wide side streams: TWO per device, whatever the number of roles.

  * "geometry" (high priority): voxelization and sparse-conv geometry, whose counts the host reads back;
  * "side" (normal priority), shared by every other role -- the momentum decoder (forward, after the encoder), the Hungarian
    assignment (loss), the bucketed gradient exchange (backward): their active phases never overlap.

Why not a stream per role.  PyTorch hands out pool streams round robin and the HIP runtime maps every NEW stream onto one of
its few hardware queues (GPU_MAX_HW_QUEUES, 8 here) in creation order.  Two streams that share a hardware queue share its
order: a stream that waits for an event of the main stream (a side stream always does) stalls everything queued behind it
on that queue -- and when that is the geometry stream, the voxel-count read-back of the NEXT step waits for the device to
reach the side stream's event: the host loses its run-ahead and the step doubles.  Measured (round 4): the assignment
kernel on a stream of its own ("matching", the fourth stream created) 61-65 ms per step against 31.4-31.6 on the momentum
decoder's stream; the bucketed gradient exchange on its own stream under a live RCCL communicator 68-71 ms
(profiles/r04_ddp_modes_hw_queues.txt).  Earlier (round 3): the side stream of the THIRD model built in a process landed
on the hardware queue of the main stream and the step ran 2 ms slower (scripts/ubench/trainer_sequence.py).  `Trainer`
creates both streams before anything else asks the pool for one, so their queues are the same in every process."""
import torch

_streams = {}
_PHYSICAL = {"geometry": ("geometry", -1)}   # every other role -> ("side", 0)


def side_stream(device, role, priority=None):
    """The stream of `role` on `device` (see the module docstring; `priority` is accepted for older callers and ignored: the
    physical stream decides)."""
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    name, prio = _PHYSICAL.get(role, ("side", 0))
    key = (index, name)
    if key not in _streams:
        _streams[key] = torch.cuda.Stream(device=torch.device("cuda", index), priority=prio)
    return _streams[key]


def create_side_streams(device):
    """Both physical streams of `device`, geometry first (engine.Trainer calls this before the model runs and before the
    first collective creates RCCL's streams)."""
    device = torch.device(device)
    if device.type == "cuda":
        side_stream(device, "geometry")
        side_stream(device, "side")
