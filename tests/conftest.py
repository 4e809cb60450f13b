import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The C++ CPU restatement (oracle/cpp), built on demand.  Test infrastructure only."""
    from oracle import cpp_oracle
    cpp_oracle.lib()
    return cpp_oracle


@pytest.fixture(scope="session")
def gpu_ctx():
    from polymath_amd import api
    ctx = api.Context(0)
    yield ctx
    ctx.close()
