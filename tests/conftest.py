import json
import os
import sys

import numpy as np
import pytest

# The suite checks ROUTES (C-ABI call counters, lazy.STATS, kernel names) of forwards it repeats with one input; the implicit hipGraphs
# (utils/implicit.py: on by default, replay from the third identical call) would hide exactly those calls.  The suite therefore runs
# with the default switched off; tests/test_gpu_r6.py switches it on for the tests of the feature itself.
os.environ.setdefault("QT_AUTO_GRAPH", "0")

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Vectors produced by running the reference (tests/golden/make_golden.py)."""
    return np.load(os.path.join(GOLDEN_DIR, "golden_v1.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_r2():
    """Round-2 vectors (tests/golden/make_golden_r2.py): XNOR activation op, deprecated functional forms."""
    return np.load(os.path.join(GOLDEN_DIR, "golden_r2_v1.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_hashes_r2():
    with open(os.path.join(GOLDEN_DIR, "golden_hashes_r2.json")) as fh:
        return json.load(fh)["cases"]


@pytest.fixture(scope="session")
def golden_hashes():
    with open(os.path.join(GOLDEN_DIR, "golden_hashes.json")) as fh:
        return json.load(fh)["cases"]


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture()
def exact_split():
    """Tests that pin the EXACT three-term bf16 route of real-valued operands (kernel names, bit-level layout); the default
    since round 3 is the two-term fp16 split (ops.FLOAT_SPLIT = "f16x2", covered by tests/test_gpu_r3.py)."""
    from pytorch_quantize_impls_amd import ops
    with ops.float_split("bf16x3"):
        yield


def same(a, b):
    """Bitwise-style equality for fp32 arrays, treating NaN == NaN."""
    a = np.asarray(a)
    b = np.asarray(b)
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def norm_err(a, b):
    """max|a-b| / max|b| — the float-tail parity metric of SURVEY.md section 8d."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


@pytest.fixture()
def s2d_first_layer():
    """Tests that pin the round-3 route of strided real-valued first layers (space-to-depth pack + implicit GEMM: kernel names,
    both float splits); since round 4 such layers take the direct kernel (ops.FIRST_DIRECT, tests/test_gpu_r4.py)."""
    from pytorch_quantize_impls_amd import ops
    prev = ops.FIRST_DIRECT
    ops.FIRST_DIRECT = False
    yield
    ops.FIRST_DIRECT = prev
