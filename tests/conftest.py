import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """The in-tree HIP library, ALWAYS brought up to date first (build() is incremental by content hash and verifies that
    the loaded library reports the hash of the current sources), never replaced by a fallback.  Round 1 built only when the
    .so was missing, so an edited kernel could be tested against yesterday's binary."""
    import __graft_entry__
    __graft_entry__.build()
    from tokenhmr_amd import _cabi
    return _cabi.load()


@pytest.fixture(scope="session")
def cuda_dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible (gpu-marked tests run on the MI355X box)")
    return torch.device("cuda:0")


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
