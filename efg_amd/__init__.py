"""MI355X-native Voxel-DETR / ConQueR training path."""
import os

# The step uses three device queues at once: the main stream, the high-priority geometry stream (voxelization and
# sparse-conv site counts, whose read-backs the host waits on) and, with more than one rank, RCCL's streams.  The
# HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4); once the RCCL
# communicator exists, the geometry stream ends up sharing a hardware queue with other work and its kernels wait
# behind unrelated ones: +1.4 ms/step with nothing else changed (scripts/ubench/ddp_modes.py none vs none:comm:
# 35.4 -> 36.8 ms; 35.8 / 35.8 with 8 queues).  Read by the runtime when it initialises, so it has to be in the
# environment before the first HIP call; an explicit setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
