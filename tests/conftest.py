import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu test selected but no GPU is visible")
    from disn_amd.api import Context
    c = Context(0)
    yield c
    c.close()
