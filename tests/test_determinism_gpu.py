"""GPU: two runs of the benchmark's own training step give identical bits (EFG_DETERMINISTIC=1).

Since round 4 nothing on the path accumulates in arrival order: the encoder's box-attention backward runs its query
tiles in colour classes with a plain read-modify-write flush, binned corners are summed as exact fixed-point integers,
the proposal top-k breaks ties by index (csrc/box_fused.hip, csrc/topk.hip); EFG_DETERMINISTIC=1 additionally takes the
one dense 3 x 3 convolution off MIOpen -- whose solver is picked by timing, so on a loaded box even its forward could be
an atomic one -- and runs it as fixed-order GEMMs over the padded channels-last map (operators/conv2d.py).
Two trainers with one seed then agree in EVERY loss term and EVERY gradient over consecutive optimizer steps."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_two_runs_of_the_full_size_step_are_bit_identical(dev):
    env = dict(os.environ, EFG_DETERMINISTIC="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ubench", "determinism_probe.py"), "--steps", "2"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    out = r.stdout
    assert r.returncode == 0, out[-2000:] + r.stderr[-3000:]
    assert "loss terms that differ: 0 of" in out, out[-3000:]
    assert "gradients that differ: 0 of" in out, out[-3000:]
    assert "step 1: total loss same, sum |grad| same" in out, out[-3000:]
