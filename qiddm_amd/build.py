"""Build the in-tree HIP extension: ``python -m qiddm_amd.build``.

Cross-compiles for gfx950 with hipcc (no GPU needed).  Output:
``qiddm_amd/lib/libqiddm_hip.so`` -- git-ignored, but it travels to the GPU box
with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libqiddm_hip.so")
SOURCES = ["qiddm_capi.hip", "qiddm_train.hip", "qiddm_qconv.hip", "qiddm_mixed.hip", "qiddm_norm.hip",
           "qiddm_wide.hip", "qiddm_cz10.hip"]
OBJ_DIR = os.path.join(LIB_DIR, "obj")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC"]


def _headers():
    hs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    return hs + [os.path.join(HERE, "..", "include", "qiddm_hip.h")]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected on PATH or at /opt/rocm/bin/hipcc)")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + _headers()
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    # one object per translation unit, compiled side by side, then one link
    procs = []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[qiddm_amd.build]", " ".join(cmd), flush=True)
        procs.append((cmd, obj, subprocess.Popen(cmd, cwd=CSRC)))
    objs = []
    for cmd, obj, proc in procs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        objs.append(obj)
    link = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs
    if verbose:
        print("[qiddm_amd.build]", " ".join(link), flush=True)
    subprocess.run(link, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
