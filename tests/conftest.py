import os
import sys

# the oracle's OpenMP loops are tiny in the tests: a thread per core of a 128-core GPU host
# (possibly under a CPU quota) only adds barrier overhead
os.environ.setdefault("OMP_NUM_THREADS", str(min(8, os.cpu_count() or 1)))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    # the reference tree (imported by test_reference_pipeline.py only) uses numpy calls deprecated in NumPy 2
    config.addinivalue_line("filterwarnings", "ignore:.*in1d.*:DeprecationWarning")


@pytest.fixture(scope="session")
def oracle64():
    from oracle.oracle import Oracle
    return Oracle("f64")


@pytest.fixture(scope="session")
def oracle32():
    from oracle.oracle import Oracle
    return Oracle("f32")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
