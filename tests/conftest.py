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
    """The CPU checker (oracle/), built on demand.  Test infrastructure only."""
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session")
def hip():
    """The product: libekfslam_hip.so through its C ABI.  Fails loudly if it is not built."""
    from ekf_slam_ml_amd import capi
    capi.load()
    if capi.device_count() < 1:
        pytest.fail("gpu test started without a visible HIP device")
    return capi
