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


@pytest.fixture(autouse=True)
def _gpu_ctx_options_restored(request):
    """The session's GPU context is shared: whatever pm_ctx_set_option calls a test makes are undone after it."""
    if "gpu_ctx" not in request.fixturenames:
        yield
        return
    from polymath_amd import api
    ctx = request.getfixturevalue("gpu_ctx")
    before = {k: ctx.get_option(k) for k in api.OPTIONS}
    yield
    for k, v in before.items():
        ctx.set_option(k, v)
