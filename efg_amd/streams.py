"""Process-wide side streams, one per device and role.

A model asks for "the geometry stream of its device" instead of making its own: PyTorch hands out pool streams round
robin and the HIP runtime maps them onto its few hardware queues (GPU_MAX_HW_QUEUES, 8 here) in creation order, so the
side stream of the THIRD model built in a process landed on the hardware queue of the main stream and the step ran 2 ms
slower (scripts/ubench/trainer_sequence.py: 33.0, 32.9, 35.0, 33.0 ms for four trainers in a row, with and without
empty_cache() between them).  One stream per role keeps the first model's queue assignment for every later one."""
import torch

_streams = {}


def side_stream(device, role, priority=0):
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, role)
    if key not in _streams:
        _streams[key] = torch.cuda.Stream(device=torch.device("cuda", index), priority=priority)
    return _streams[key]
