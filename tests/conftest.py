import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order of the GPU suite: the driver runs `pytest -x` inside a time limit, so a failure or a kill
# at the limit leaves every LATER test unreported.  Oracle-parity tests come first, the heavy legs (full-size configurations,
# rank processes, soak, launch chains) last; inside a tier the files' own order is kept.  CPU tests are not reordered.
_FIRST = ("test_gpu_parity.py", "test_gpu_shapes.py", "test_gpu_configs.py::test_whole_proof_bit_exact_vs_oracle_at_size")
_HEAVY = ("test_config_2p24", "test_config_2p22", "test_config_bn254_2p20", "test_full_size_2p20", "test_hundreds_of_proofs", "test_bench_",
          "test_three_contexts_prove_concurrently", "test_rccl_standin.py", "test_soak.py", "test_ntt_2p25", "test_msm_full_size",
          "test_ntt_full_size", "test_ntt_bn254_2p21")


def _tier(item):
    nid = item.nodeid
    if item.get_closest_marker("gpu") is None:
        return 1
    if any(h in nid for h in _HEAVY):
        return 3
    if any(f in nid for f in _FIRST):
        return 0
    return 2


def pytest_collection_modifyitems(config, items):
    items.sort(key=_tier)          # stable: the files' own order inside a tier


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
