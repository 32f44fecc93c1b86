import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _fresh_conv1_route(request):
    """The choice of conv1 kernels (f16 pipes / f32-input kernels for nearly constant channels: cpp_ctx_set_route_threshold) lives on the
    process-wide default context and outlasts an agent: every GPU test starts on the f16 pipes with nothing seen yet, whatever the test
    before it trained on."""
    if request.node.get_closest_marker("gpu") is not None:
        from cartpoleplusplus_amd import _lib
        if _lib._default_ctx is not None:
            ctx = _lib.default_context()
            ctx.sync()
            ctx.set_route_threshold(0.0)
            ctx.set_route_threshold(100.0)
    yield
