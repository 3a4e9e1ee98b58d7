import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle (float64 / fp32 torch on the host) is most of the GPU suite's wall time, and the GPU boxes have 256 hardware
    # threads: torch's default of one thread per hardware thread is several times SLOWER on this op mix (30 x 40 rays, 128-wide GEMMs)
    # than a moderate count -- bench.py's cpu_baseline calibrates the same thing and lands on 32.  NEFES_TEST_THREADS overrides.
    try:
        import torch
        n = int(os.environ.get("NEFES_TEST_THREADS", "0")) or min(32, os.cpu_count() or 1)
        torch.set_num_threads(n)
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return load
