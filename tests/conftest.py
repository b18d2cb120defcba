import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fx():
    """The product package (directory name has a hyphen, hence importlib)."""
    mod = importlib.import_module("gr-liquiddsp_amd")
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    return mod


@pytest.fixture(scope="session")
def oracle():
    import oracle_ffi
    oracle_ffi.lib()
    return oracle_ffi
