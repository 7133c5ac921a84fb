import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_mirt():
    """The package directory name has hyphens, so it is imported through importlib.  On a clean checkout the native
    libraries are compiled first (hipcc cross-compiles gfx950 without a GPU); the product itself never builds or falls
    back on its own — it raises if libmirt.so is missing."""
    mirt = importlib.import_module("cpu-raytracing-experiments_amd")
    if not os.path.exists(mirt.LIB_PATH):
        mirt.build()
    return mirt


@pytest.fixture(scope="session")
def mirt():
    return load_mirt()


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle_binding
    return oracle_binding.load()
