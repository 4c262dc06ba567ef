import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def hip_lib():
    """libwaxhip.so, built in-tree if missing (hipcc cross-compiles without a GPU)."""
    from wax_amd import _abi, build
    build.build()
    return _abi.lib()


def pytest_collection_modifyitems(config, items):
    """A hung GPU test must not eat the box's time budget: every gpu-marked test gets a hard timeout
    (pytest-timeout, when installed)."""
    for item in items:
        if "gpu" in item.keywords and not any(m.name == "timeout" for m in item.iter_markers()):
            item.add_marker(pytest.mark.timeout(240))
