import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in (HERE, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (gpu tests run on the MI355X box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Tests need the oracle .so and (for ABI checks) the product .so files."""
    import importlib
    pkg = importlib.import_module("gadget-leicester_amd")
    if not (os.path.exists(pkg.LIBGHIP) and os.path.exists(pkg.LIBHOST)):
        pkg.build()
    from oracle import oracle as O
    O.lib()
    yield
