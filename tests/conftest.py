import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# every fused box-attention backward checks that no binned grad_value entry overflowed its bin (a sync per call)
os.environ.setdefault("EFG_CHECK_BINS", "1")


_T0 = time.monotonic()
_CRUMBS = {"fd": None, "passed": 0, "failed": 0, "file": None, "fault": None}


def _crumb(text):
    """One unbuffered line on the process's REAL stderr (dup'ed before pytest's fd capture starts) and in
    gpurun_out/gpu_test_progress.log: a suite that dies of SIGABRT/SIGSEGV (a GPU memory fault makes ROCr call
    abort()) still leaves 'last test started = X, N passed' as the last thing in the log tail."""
    line = ("\n[efg-test +%.1fs] %s\n" % (time.monotonic() - _T0, text)).encode()
    for fd in (_CRUMBS["fd"], _CRUMBS["file"]):
        if fd is not None:
            try:
                os.write(fd, line)
            except OSError:
                pass


@pytest.hookimpl(trylast=True)
def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import faulthandler

    try:
        _CRUMBS["fd"] = os.dup(2)
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        tag = os.environ.get("EFG_TEST_LOG_TAG", "")   # a pytest started BY a test logs next to, not over, its parent
        _CRUMBS["file"] = os.open(os.path.join(out, "gpu_test_progress%s.log" % tag), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        # the interpreter's fatal-signal dump (all threads + the 3 kB extension-module list) goes to a file so
        # that it cannot push the breadcrumbs out of the log tail a driver keeps
        _CRUMBS["fault"] = open(os.path.join(out, "gpu_test_faulthandler%s.log" % tag), "w")
        faulthandler.enable(file=_CRUMBS["fault"], all_threads=True)
    except OSError:
        faulthandler.enable(all_threads=True)


def pytest_runtest_logstart(nodeid, location):
    _crumb("%d passed, %d failed; START %s" % (_CRUMBS["passed"], _CRUMBS["failed"], nodeid))


def pytest_runtest_logreport(report):
    if report.when == "call":
        if report.passed:
            _CRUMBS["passed"] += 1
        elif report.failed:
            _CRUMBS["failed"] += 1
            _crumb("FAILED %s" % report.nodeid)
    elif report.failed:
        _CRUMBS["failed"] += 1
        _crumb("ERROR(%s) %s" % (report.when, report.nodeid))


def pytest_sessionfinish(session, exitstatus):
    _crumb("SESSION FINISHED exit=%s: %d passed, %d failed" % (exitstatus, _CRUMBS["passed"], _CRUMBS["failed"]))


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


def force_proposals(transformer, indexes):
    """Test-side hook: make `transformer` take the given proposal tokens instead of its own top-k.  WHICH tokens make
    the cut is ill-defined when the k-th best score is shared by many tokens (empty BEV cells all produce the same logit;
    on a random-init model the cut often falls inside that plateau, and two devices whose logits differ in the last bit
    then pick different members of it).  A comparison feeds both sides one set of proposals; the tests check separately
    that the two sides' own sets differ only inside the tie.  `indexes` None restores the model's own selection."""
    import torch

    if indexes is None:
        transformer.__dict__.pop("_select_proposals", None)
        return
    transformer._select_proposals = lambda probs: (torch.gather(probs, 1, indexes.to(probs.device)), indexes.to(probs.device))
