import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# a HIP-graph capture that silently falls back to eager launches would hide a regression of the host-side path
os.environ.setdefault("EFG_GT_GRAPH_STRICT", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(autouse=True)
def _restore_gc():
    """A Trainer that runs a step on the GPU freezes and disables the cyclic collector (engine.py); give the
    next test the interpreter's default back."""
    import gc

    yield
    if not gc.isenabled():
        gc.enable()
        gc.unfreeze()
        gc.collect()


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU checker (oracle/).  Test infrastructure only."""
    import oracle

    oracle.build(with_ref=True)
    return oracle


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
