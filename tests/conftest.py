import gzip
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rl():
    """The product package (directory `rendering-learning_amd`)."""
    return importlib.import_module("rendering-learning_amd")


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    return importlib.import_module("rl_oracle")


def read_golden(name):
    p = os.path.join(GOLDEN, name)
    if name.endswith(".gz"):
        with gzip.open(p, "rb") as f:
            return f.read()
    with open(p, "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def golden():
    return read_golden
