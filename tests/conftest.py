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
    """The in-tree HIP library; built on demand where hipcc exists, never replaced by a fallback."""
    from tokenhmr_amd import _cabi
    if not os.path.exists(_cabi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _cabi.load()


@pytest.fixture(scope="session")
def cuda_dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible (gpu-marked tests run on the MI355X box)")
    return torch.device("cuda:0")


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
