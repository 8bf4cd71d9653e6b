import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session", autouse=True)
def _fresh_library():
    """Rebuild libeemflow_hip.so when a source is newer (no-op otherwise; needs hipcc, skipped without)."""
    import shutil
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        from eemflow_amd.build import build_library
        build_library(verbose=False)
    yield
