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


# Collection order of the GPU suite: oracle / fixture PARITY first, plumbing last.  The driver runs ``pytest tests/ -x``: whatever
# fails first hides everything behind it (round 3: an abort in the third file blanked 417 parity tests), so the files whose
# tests are the parity evidence go first and everything that spawns processes or drives whole entry points goes last.
# (Tests that need a process group run in child processes anyway: tests/dist_child.py.)
_ORDER = ["test_gpu_kernels", "test_gpu_models", "test_gpu_configs", "test_gpu_blockwise", "test_gpu_precise", "test_gpu_fullsize",
          "test_gpu_meters", "test_gpu_feature_store", "test_gpu_step_structures", "test_gpu_entrypoints", "test_gpu_metric_target",
          "test_gpu_two_rank", "test_gpu_dist"]


def pytest_collection_modifyitems(session, config, items):
    if os.environ.get("EGK_TEST_KEEP_ORDER"):  # (reproductions of an order-dependent fault: tools/round4/repro_abort.sh)
        return
    def rank(item):
        stem = Path(str(item.fspath)).stem
        if stem in _ORDER:
            return (1, _ORDER.index(stem))
        if item.get_closest_marker("gpu") is None:
            return (0, 0)  # CPU tests first
        return (1, len(_ORDER) - 2.5)  # GPU files this list does not know: after the parity files, before the process spawners
    items.sort(key=rank)  # (stable: the order inside a file is kept)


def _install_abort_backtrace():
    """EGK_ABORT_BT=<path of tools/round4/abort_bt.so>: print the C backtrace of a thread that raises SIGABRT (diagnostics)."""
    so = os.environ.get("EGK_ABORT_BT")
    if so and Path(so).exists():
        import ctypes
        import faulthandler
        faulthandler.enable(all_threads=True)
        ctypes.CDLL(so).abort_bt_install()


def pytest_sessionstart(session):
    _install_abort_backtrace()
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
