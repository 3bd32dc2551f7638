import os
import sys
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bmsparse-spgemm-spmv_amd"))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "tests"))

GOLDEN = os.path.join(REPO, "tests", "golden")
MTX = os.path.join(GOLDEN, "mtx")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU device node (/dev/kfd) in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def _ensure_built():
    """the built libraries travel with the tree; if a checkout arrives without them (fresh clone), build once."""
    need = [os.path.join(REPO, "bmsparse-spgemm-spmv_amd", "lib", "libbmsp.so"), os.path.join(REPO, "oracle", "libbmsp_oracle.so")]
    if all(os.path.exists(p) for p in need):
        return
    sys.path.insert(0, REPO)
    import __graft_entry__
    __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    _ensure_built()
    import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def bmsp():
    """the product binding; raises (never falls back) if libbmsp.so is missing."""
    _ensure_built()
    import pybmsp
    pybmsp.lib()
    return pybmsp
