"""Demonstrates the abort behind GPUTEST_r02 (rc 134): a dead reference cycle that owns a CUDAGraph is collected while
ANOTHER capture is under way; on ROCm ~CUDAGraph calls hipDeviceSynchronize under AT_CUDA_CHECK, which is illegal inside
a capture, the check throws in a destructor and the process aborts.  Expected: this script dies with SIGABRT (rc 134)
with plain torch.cuda.graph; efg_amd.hipgraph.capture avoids it (tests/test_gt_graph_gpu.py)."""
import gc

import torch

x = torch.zeros(1024, device="cuda")


class Holder:
    pass


g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    y = x + 1
h = Holder()
h.graph, h.me = g1, h
del g1, h
print("cycle with a captured graph is dead; capturing another graph and collecting inside it", flush=True)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, capture_error_mode="thread_local"):
    z = x * 2
    gc.collect()
print("SURVIVED (no abort)", flush=True)
