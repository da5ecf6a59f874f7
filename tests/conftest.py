import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_sessionstart(session):
    # PyTorch-ROCm wheels carry their own HIP runtime: when both runtimes end up in one process, torch's has to claim the GPU
    # first.  Some GPU tests use torch as the float32 cross-check of the conv stack, so do that up front,
    # whatever order the test files run in.
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # noqa: BLE001 -- no torch, or no GPU: nothing to order
        pass


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle

    oracle.lib()
    return oracle
