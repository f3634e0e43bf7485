import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))

    return load


@pytest.fixture(autouse=True, scope="module")
def _release_gpu_memory_between_modules():
    """A test module that built a full-size workload (tests/test_config_shapes.py: c3 / c5 as bench.py builds them) leaves tens
    of GB cached in this process's allocator; modules that run their case in a CHILD process (race check, multi-rank, CLI) then
    find the card full.  Hand the cache back after every module."""
    yield
    if "torch" in sys.modules:
        import gc
        import torch
        gc.collect()
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.empty_cache()
