import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests skip (not fail) on a box without a HIP device, e.g. a plain `pytest` in the build container"""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no HIP device here)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import gxo
    gxo.lib()
    return gxo
