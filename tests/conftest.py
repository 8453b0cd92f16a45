import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def engine():
    """The process-wide engine on cuda:0.  GPU tests must exercise libvdb.so: a missing library
    or device is a failure, never a skip to some other implementation."""
    from vectordb_amd.engine import Engine
    if not _gpu_available():
        pytest.fail("GPU test selected but no GPU is visible")
    return Engine(device=0)


_counter = [0]


@pytest.fixture
def table_name(engine):
    _counter[0] += 1
    name = "t%d" % _counter[0]
    yield name
    if engine.has_table(name):
        engine.drop_table(name)
