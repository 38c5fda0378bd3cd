"""pytest configuration: `-m gpu` tests need a real MI355X and call through the C ABI; everything else runs on CPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (llama-gguf_amd/), with its native libraries built."""
    lib = os.path.join(graft.PKG_DIR, "lib", "libllama_gguf_hip.so")
    if not os.path.exists(lib):
        graft.build()
    return graft.load_package()


@pytest.fixture(scope="session")
def orc():
    """The CPU parity oracle (test infrastructure only)."""
    return graft.load_oracle()


@pytest.fixture(scope="session")
def gpu(pkg):
    """Fails (does not skip) when a `-m gpu` test runs without the engine or without a device."""
    n = pkg.hip_backend.device_count()
    assert n >= 1, "GPU test selected but no HIP device is visible"
    return pkg.hip_backend
