import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A checkout without the built library (the .so is git-ignored) compiles it once, in-tree, when hipcc is around:
    the tests exercise the product path, and the product path itself never builds or falls back on its own."""
    import shutil
    lib = REPO / "egopack_amd" / "libegopack_hip.so"
    if not lib.exists() and (shutil.which("hipcc") or Path("/opt/rocm/bin/hipcc").exists()):
        from egopack_amd import build
        build.build_library(force=False, verbose=False)


@pytest.fixture(scope="session")
def golden():
    import torch

    def load(name):
        return torch.load(GOLDEN / f"{name}.pt", map_location="cpu", weights_only=False)

    return load
