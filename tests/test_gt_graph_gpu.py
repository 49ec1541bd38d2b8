"""The momentum ("GT") decoder replayed from a HIP graph gives the same outputs as the eager pass."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gt_decoder_graph_matches_eager(monkeypatch):
    from efg_amd.engine import Trainer

    # here (and only here) a capture that silently falls back to eager launches is a failure
    monkeypatch.setenv("EFG_GT_GRAPH_STRICT", "1")
    tr = Trainer(device="cuda:0", overrides={"model.transformer.num_queries": 60}, seed=0)
    t = tr.model.transformer
    g = torch.Generator(device="cuda").manual_seed(0)
    memory = torch.randn(2, 188 * 188, 256, device="cuda", generator=g) * 0.1
    shape = torch.tensor([[188, 188]], device="cuda")
    start = torch.zeros(1, dtype=torch.int64, device="cuda")
    for n in (48, 96, 48):  # two shapes, the first one twice (capture, capture, replay)
        props = torch.rand(2, n, 10, device="cuda", generator=g) * 0.8 + 0.1
        grp = torch.arange(n, device="cuda") // 16
        mask = grp[:, None] != grp[None, :]
        with torch.no_grad():
            monkeypatch.setenv("EFG_GT_GRAPH", "0")
            eager = t._run_gt_decoder(memory, shape, start, props, mask)
            monkeypatch.setenv("EFG_GT_GRAPH", "1")
            graphed = t._run_gt_decoder(memory, shape, start, props, mask)
            graphed = [x.clone() for x in graphed]  # static buffers: copy before the next replay
        assert not getattr(t, "_gt_graph_off", False)
        for a, b in zip(eager, graphed):
            assert a.shape == b.shape
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), float((a - b).abs().max())
    assert len(t._gt_graphs) == 2 and t._gt_graph_stats == [1, 2]
    # with autograd on (not the momentum decoder's situation) the eager path is taken: nothing new is captured
    t._run_gt_decoder(memory, shape, start, props, mask)
    assert t._gt_graph_stats == [1, 2]


def _run_child(code):
    import os
    import subprocess
    import sys

    from conftest import ROOT

    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT,
                          env=dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")))


_CYCLE_WITH_GRAPH = """
import gc, torch
from efg_amd.hipgraph import capture, CaptureFailed
x = torch.zeros(1024, device="cuda")
class Holder: pass
def make_dead_cycle():                 # what a dropped model leaves behind: a reference cycle that owns a captured graph
    g, _ = capture(lambda: x + 1, "cuda:0")
    h = Holder(); h.graph = g; h.me = h
gc.disable()
make_dead_cycle()
"""


def test_capture_collects_dead_graphs_before_not_during():
    """A dead reference cycle owning an older graph must be destroyed BEFORE the next capture starts and the collector
    must be off during it: on ROCm ~CUDAGraph synchronises the device, which inside a capture throws from a destructor
    and aborts the process (GPUTEST_r02: rc 134).  In a child process, so that a regression is a named failure."""
    r = _run_child(_CYCLE_WITH_GRAPH + """
seen = []
def region():
    seen.append(gc.isenabled())
    gc.collect()                       # stands for an automatic collection falling into the captured region
    return x * 2
gc.enable()
g, y = capture(region, "cuda:0")
assert seen and not any(seen), seen    # collector off during warm-up and capture
assert gc.isenabled()                  # and restored afterwards
g.replay(); torch.cuda.synchronize()
assert float(y.sum()) == 0.0
print("CHILD_OK")
""")
    assert r.returncode == 0 and "CHILD_OK" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])


def test_failed_capture_restores_the_stream_and_raises_capture_failed():
    r = _run_child("""
import torch
from efg_amd.hipgraph import capture, CaptureFailed
torch.cuda.manual_seed(77); torch.rand(5, device="cuda")
x = torch.zeros(1024, device="cuda")
cur = torch.cuda.current_stream()
calls = []
def region():
    calls.append(1)
    if len(calls) == 3:                # warm-up passes are fine, the captured pass is not
        return float(x.sum())          # a D2H read-back is illegal inside a capture
    return x + 1
try:
    capture(region, "cuda:0")
    raise SystemExit("capture of an illegal region did not fail")
except CaptureFailed as exc:
    assert exc.__cause__ is not None
assert torch.cuda.current_stream() == cur          # not left on the capture stream
assert not torch.cuda.is_current_stream_capturing()
r = torch.rand(8, device="cuda")                   # the default generator is not left in its "capturing" state ...
torch.cuda.manual_seed(77); torch.rand(5, device="cuda"); want = torch.rand(8, device="cuda")
assert torch.equal(r, want), (r, want)             # ... and goes on where it was (seeded 77 + 5 draws by the prelude)
y = (x + 3).sum().item()                           # the device is usable, eagerly ...
g, z = capture(lambda: x + 2 + 0 * torch.rand(1024, device="cuda"), "cuda:0")   # ... and for the next capture
g.replay(); torch.cuda.synchronize()
assert y == 3 * 1024 and float(z.sum()) == 2 * 1024
# ROCm leaves an invalidated capture stream invalidated for good: it must not be one of PyTorch's pool streams, which
# come round again every 32 requests
for _ in range(70):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        w = x + 1
    s.synchronize()
# a region that fails in Python (nothing illegal for HIP) ends its capture normally; same contract for the caller
def bad():
    if len(calls) >= 3:
        calls.append(1)
        if len(calls) == 7:
            raise ValueError("shape mismatch, say")
    return x + 1
calls[:] = [1, 1, 1, 1]
try:
    capture(bad, "cuda:0")
    raise SystemExit("capture of a raising region did not fail")
except CaptureFailed as exc:
    assert isinstance(exc.__cause__, ValueError)
assert torch.cuda.current_stream() == cur and torch.rand(2, device="cuda").numel() == 2
g2, z2 = capture(lambda: x + 5, "cuda:0")
g2.replay(); torch.cuda.synchronize()
assert float(z2.sum()) == 5 * 1024
print("CHILD_OK")
""")
    assert r.returncode == 0 and "CHILD_OK" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])


def test_momentum_decoder_on_its_own_stream_changes_no_bit(monkeypatch):
    """EFG_GT_STREAM: the momentum decoder issued right after the encoder on a side stream (beside the decoder) or on the
    main stream after it -- the same kernels on the same inputs, so three training steps give the same loss terms and the
    same parameters bit for bit (the EMA update of the momentum decoder reads the optimizer's output of the previous
    step on the OTHER stream: a missing dependency would show here)."""
    from efg_amd.engine import Trainer, synthetic_batch

    dev = torch.device("cuda:0")
    # the benchmark's own size with EFG_DETERMINISTIC=1: there two runs of a step agree bit for bit (tests/test_determinism_gpu.py;
    # at reduced shapes the GEMM library picks atomic split-K solutions), so any difference here is the stream's
    monkeypatch.setenv("EFG_DETERMINISTIC", "1")
    runs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("EFG_GT_STREAM", flag)
        tr = Trainer(device=dev, seed=0)
        tr.model.noise_generator = torch.Generator().manual_seed(77)
        trace = []
        for s in range(3):
            losses, total = tr.step(synthetic_batch(900 + s, 2, device=dev))
            trace.append({k: float(v.detach()) for k, v in losses.items()})
        torch.cuda.synchronize()
        runs.append((trace, [p.detach().clone() for p in tr.model.parameters()]))
        tr.close()
        del tr
    (ta, pa), (tb, pb) = runs
    assert ta == tb
    assert all(torch.equal(a, b) for a, b in zip(pa, pb))
