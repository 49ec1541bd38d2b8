"""Which clean-up after a failed capture leaves torch usable?  Each variant runs in a child process (GPU box).

Findings (torch 2.10.0+rocm7.0, profiles/r03_failed_capture_variants.txt): capture_end() of an invalidated capture throws
before the generators' capture_epilogue(), so the device's default generator keeps capturing_=true and every later
random op raises "Offset increment outside graph capture encountered unexpectedly"; reset() / deleting the graph do not
clear it, and a second ("healing") capture cannot even begin.  Swapping the generator's state object for its clone
(seed and offset kept, capturing_ false) is what works -- efg_amd/hipgraph.py `_restore_generator`."""
import subprocess
import sys

CHILD = r'''
import sys, gc, torch
variant = sys.argv[1]
torch.cuda.manual_seed(1234)
torch.rand(7, device="cuda")
x = torch.zeros(1024, device="cuda")
side = torch.cuda.Stream()
graph = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.stream(side):
    graph.capture_begin(capture_error_mode="thread_local")
    try:
        float(x.sum())
    except Exception as e:
        print("in-capture failure:", str(e)[:40])
    try:
        graph.capture_end()
    except Exception as e:
        print("capture_end:", str(e)[:40])
if "reset" in variant:
    graph.reset()
if "del" in variant:
    del graph
    gc.collect()
    print("deleted")
if "heal" in variant:
    g = torch.cuda.CUDAGraph()
    t = torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    try:
        with torch.cuda.stream(side):
            g.capture_begin(capture_error_mode="thread_local")
            t.add_(0)
            g.capture_end()
        print("healed")
    except Exception as e:
        print("heal failed:", str(e)[:200].replace("\n", " "))
        import os; sys.stdout.flush(); os._exit(0)
if "clone" in variant:
    gen = torch.cuda.default_generators[0]
    before = (gen.initial_seed(), gen.get_offset() if "offs" in variant else None)
    gen.graphsafe_set_state(gen.clone_state())
    print("cloned, seed kept:", gen.initial_seed() == 1234)
try:
    r = torch.rand(4, device="cuda"); torch.cuda.synchronize()
    print("RNG OK")
except Exception as e:
    print("RNG BROKEN:", str(e)[:60])
if "clone" in variant:
    # the stream of numbers continues where it was: same as a process that never tried to capture
    torch.cuda.manual_seed(1234); torch.rand(7, device="cuda"); ref = torch.rand(4, device="cuda")
    print("continues the sequence:", bool((ref == r).all()))
    # and later captures with random ops in them work
    g2 = torch.cuda.CUDAGraph(); out = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
    with torch.cuda.stream(torch.cuda.Stream()):   # NOT `side`: that one stays invalidated (invalidated_stream.py)
        g2.capture_begin(capture_error_mode="thread_local")
        out.copy_(torch.rand(4, device="cuda"))
        g2.capture_end()
    g2.replay(); a = out.clone(); g2.replay(); b = out.clone(); torch.cuda.synchronize()
    print("graph rng replays differ:", bool((a != b).any()), "eager after:", torch.rand(2, device="cuda").numel() == 2)
    del g2
if "late" in variant:
    del graph
    gc.collect()
    print("late delete ok")
print("END")
'''

for v in ["none", "reset", "reset+del", "reset+heal", "reset+clone", "reset+clone+late", "reset+del+clone"]:
    r = subprocess.run([sys.executable, "-c", CHILD, v], capture_output=True, text=True)
    err = [l for l in r.stderr.splitlines() if "what()" in l or "Error" in l]
    print("%-18s rc=%4d | %s | %s" % (v, r.returncode, " ; ".join(l for l in r.stdout.strip().splitlines() if l), err[-1:]))
