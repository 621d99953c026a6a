import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only), built on demand."""
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def ca():
    """The product package; its shared library must already be built."""
    import compose_amd
    compose_amd.lib()
    return compose_amd


@pytest.fixture(scope="session")
def gpu(ca):
    # Some GPU tests use torch (device tensors, bench.py): import it before the library resolves
    # librccl, so that one process never ends up with two different copies of RCCL.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if ca.device_count() < 1:
        pytest.fail("no GPU visible: -m gpu tests must run on the GPU box")
    return ca
