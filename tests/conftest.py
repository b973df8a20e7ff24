import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    lib = os.path.join(ROOT, "yahr_amd", "libyahr_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session", autouse=True)
def built():
    _ensure_built()
    import oracle_py
    oracle_py.build_oracle()
    yield


def has_gpu():
    import yahr_amd
    return yahr_amd.device_count() > 0


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run the HIP path: no skip-on-missing-device, the C ABI fails loudly instead."""
    import yahr_amd
    n = yahr_amd.device_count()
    assert n > 0, "no HIP device visible: -m gpu tests need a real MI355X (there is no CPU fallback)"
    return n
