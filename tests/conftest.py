import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# compiled plugins of the test suite are cached inside the repository tree (the library's default, ~/.cache/ocs_amd, lies outside)
os.environ.setdefault("OCS_JIT_CACHE_DIR", os.path.join(ROOT, ".pytest_cache", "ocs_jit"))
os.makedirs(os.path.join(ROOT, ".pytest_cache"), exist_ok=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): oracle/ocs_oracle.c through ctypes."""
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def test_problem_params():
    # tests/solve_test_problem.m:11-16
    return {"c": 1.5, "m": 3.0, "r": 0.05}
