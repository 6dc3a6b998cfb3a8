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
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests need a HIP device: on a host without one they are SKIPPED (not failed), so a plain
    `pytest tests` on a CPU box reports only real CPU regressions."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a HIP GPU (MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
