"""CPU: pin the oracle's box / deformable attention against golden vectors produced from the
reference's ms_deform_attn_core_pytorch (efg/operators/ms_deform_attn.py:55-76) and autograd."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden

CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "msda_*.npz")))


@pytest.mark.parametrize("case", CASES)
def test_oracle_forward(oracle_mod, case):
    g = golden(case)
    out = oracle_mod.msda_forward(g["value"], g["shapes"], g["level_start"], g["loc"], g["attn"])
    np.testing.assert_allclose(out, g["out_fp64"], rtol=1e-5, atol=1e-5)  # tolerance: fp32 roundoff
    np.testing.assert_allclose(out, g["out_fp32"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("case", CASES)
def test_oracle_backward(oracle_mod, case):
    g = golden(case)
    gv, gl, ga = oracle_mod.msda_backward(g["value"], g["shapes"], g["level_start"], g["loc"], g["attn"],
                                          g["grad_out"])
    np.testing.assert_allclose(gv, g["grad_value"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ga, g["grad_attn"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gl, g["grad_loc"], rtol=1e-4, atol=2e-4)
