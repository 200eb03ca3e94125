import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "survey_known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gpu_ops():
    """The HIP path.  Fails (never skips, never falls back) when run without the extension."""
    import torch
    from mlmcpathintegral_amd import abi, ops
    abi.load()
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    torch.cuda.set_device(0)
    return ops
