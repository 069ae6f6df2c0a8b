import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
    # Multi-rank GPU tests start their ranks from a fork server.  It has to exist before this
    # process initialises the GPU: its children are plain forks of a process that never held a
    # HIP context, and no program is ever started from a process that does.
    try:
        import multiprocessing as mp
        import multiprocessing.forkserver as fs
        mp.set_forkserver_preload([])
        fs.ensure_running()
    except Exception:
        pass


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden
