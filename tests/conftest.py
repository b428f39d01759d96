import os
import sys

import pytest
import torch  # noqa: F401  -- FIRST: torch must load its bundled HIP runtime before libtriton_hip.so pulls in the
#                system one; a process that initialises the system runtime first makes torch.cuda report
#                "No HIP GPUs are available" (tests/test_sharded_prover.py shares a process with the other gpu tests)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/), built on demand."""
    from oracle import oracle

    oracle.build()
    return oracle


def _make_ctx(kind):
    if kind == "emu":
        from tests.emu_fixture import emu_context

        return emu_context()
    # the product library, built by hipcc, on a real MI355X -- no fallback
    from triton_vm_amd import Context

    return Context(device=0)


@pytest.fixture(scope="module", params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def ctx(request):
    """The C ABI behind either the TEST-ONLY CPU fiber emulation of the kernel sources ("emu", runs
    in the GPU-less container) or the real hipcc-built libtriton_hip.so on cuda:0 ("gpu")."""
    c = _make_ctx(request.param)
    c.kind = request.param
    yield c
    c.close()
