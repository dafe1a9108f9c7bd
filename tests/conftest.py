import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# MIOpen "fast" find mode for the TEST processes (and the children they start): with torch.backends.cudnn.benchmark the library times every
# candidate solver for each new convolution shape -- the tests that build whole trainers at odd image sizes spent 400 of the GPU suite's
# 620 s there (round 6: 139 -> 12 s, 130 -> 65 s, 96 -> 18 s, 40 -> 9 s for the four slowest).  The tests check results against oracles /
# goldens with tolerances that do not depend on which library solver ran; bench.py and the trainers are not affected (they never see this).
os.environ.setdefault("MIOPEN_FIND_MODE", "2")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests must not silently pass on a machine without a GPU
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container (gpu-marked tests run on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import numpy as np

    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
