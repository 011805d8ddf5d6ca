import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tf-mpc_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the full-size version of a test whose sampled version runs by default (TFMPC_SLOW=1 runs it)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a GPU skips the `gpu`-marked tests instead of failing them (the product
    path raises there: no CPU fallback).  `torch.cuda.device_count()` does not initialise the GPU."""
    import torch
    if os.environ.get("TFMPC_SLOW") != "1":           # the exhaustive versions (>= 1 000-case fuzz, every-pass teacher forcing) are opt-in
        slow = pytest.mark.skip(reason="full-size version: TFMPC_SLOW=1 (a sampled version of the same test runs by default)")
        for item in items:
            if "slow" in item.keywords:
                item.add_marker(slow)
    if torch.cuda.device_count() > 0:
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no ROCm device visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load
