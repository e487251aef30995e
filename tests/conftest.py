import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _usable_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup quota (a GPU box shows all of the host's cores
    to a job that may use 16 of them)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


# The C oracle (oracle/libtzoracle.so, OpenMP) reads these when it loads: one thread per USABLE core and no spinning --
# with one thread per visible core the oracle's small loops spend their time waiting for each other (measured on the GPU
# box: a 64x96 DWP rollout of 10 frames 18 s against 1.1 s).
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, _usable_cpus())))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
