"""Build libegopack_hip.so (gfx950) from egopack_amd/csrc/*.hip with hipcc, in-tree.

hipcc cross-compiles without a GPU.  The built library sits next to this file so that it travels
with a repo snapshot; it is git-ignored (source-only history)."""
from __future__ import annotations

import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OBJ = CSRC / "build"
LIB = HERE / "libegopack_hip.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: cannot build libegopack_hip.so")
    return exe


def sources():
    return sorted(CSRC.glob("*.hip"))


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> Path:
    hipcc = _hipcc()
    OBJ.mkdir(parents=True, exist_ok=True)
    headers = list(CSRC.glob("*.h")) + [HERE.parent / "include" / "egopack_hip.h"]
    jobs = []
    for src in sources():
        obj = OBJ / (src.stem + ".o")
        if force or _stale(obj, [src, *headers]):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, *FLAGS, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stderr}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        for warn in ex.map(compile_one, jobs):
            if warn and verbose:
                print(warn, file=sys.stderr)
    objs = [OBJ / (s.stem + ".o") for s in sources()]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
