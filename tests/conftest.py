"""pytest wiring: `gpu` marker, import paths for the oracle (test infrastructure) and the package."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """The package directory is named after the reference repo (it contains dashes), so it is
    imported by name through importlib."""
    return importlib.import_module("prgs-sdr-kspecanal_amd")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def ksa():
    return load_pkg()
