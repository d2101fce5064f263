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
           "qiddm_wide.hip", "qiddm_cz10.hip", "qiddm_lean.hip"]
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


_INC = __import__("re").compile(r'^\s*#\s*include\s+"([^"]+)"', __import__("re").M)


def _deps(path: str, seen=None) -> set:
    """`path` and every file it includes with quotes, transitively (the translation units share headers unevenly:
    a change to one kernel header should not rebuild all of them)."""
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path) as f:
        text = f.read()
    for inc in _INC.findall(text):
        _deps(os.path.join(os.path.dirname(path), inc), seen)
    return seen


def _obj_of(src: str) -> str:
    return os.path.join(OBJ_DIR, src.replace(".hip", ".o"))


def _stale_sources() -> list:
    out = []
    for src in SOURCES:
        obj = _obj_of(src)
        if not os.path.exists(obj):
            out.append(src)
            continue
        t = os.path.getmtime(obj)
        if any(os.path.getmtime(d) > t for d in _deps(os.path.join(CSRC, src))):
            out.append(src)
    return out


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return bool(_stale_sources()) or any(os.path.getmtime(_obj_of(s)) > t for s in SOURCES)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    # one object per translation unit whose sources changed, compiled side by side, then one link
    procs = []
    for src in (SOURCES if force else _stale_sources()):
        obj = _obj_of(src)
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[qiddm_amd.build]", " ".join(cmd), flush=True)
        procs.append((cmd, obj, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, obj, proc in procs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    objs = [_obj_of(src) for src in SOURCES]
    link = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs
    if verbose:
        print("[qiddm_amd.build]", " ".join(link), flush=True)
    subprocess.run(link, check=True, cwd=CSRC)
    return LIB


ASAN_LIB = os.path.join(LIB_DIR, "libqiddm_hip_asan.so")


def asan_runtime() -> str:
    """The shared AddressSanitizer runtime of the ROCm clang (LD_PRELOAD it into a python that loads ASAN_LIB)."""
    out = subprocess.run([_hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    path = out.stdout.strip()
    if not os.path.isabs(path) or not os.path.exists(path):
        import glob
        hits = sorted(glob.glob("/opt/rocm*/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
        if not hits:
            raise RuntimeError("libclang_rt.asan-x86_64.so not found under /opt/rocm")
        path = hits[-1]
    return path


def build_asan(verbose: bool = False) -> str:
    """HOST side of the C ABI under AddressSanitizer (CPU box only: GPU ASan / xnack+ is not available on this pool).
    `--cuda-host-only` compiles the entry points, their argument validation and the launch wrappers without any device
    code (seconds); kernels cannot be launched from this build -- it exists for tests/test_capi_asan.py."""
    obj_dir = os.path.join(LIB_DIR, "obj_asan")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    flags = ["-O1", "-g", "-std=c++17", "--offload-arch=gfx950", "--cuda-host-only", "-fsanitize=address",
             "-fno-omit-frame-pointer", "-fPIC"]
    procs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[qiddm_amd.build]", " ".join(cmd), flush=True)
        procs.append((cmd, obj, subprocess.Popen(cmd, cwd=CSRC)))
    objs = []
    for cmd, obj, proc in procs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        objs.append(obj)
    # a host-only object refers to its translation unit's device image by an external symbol (`__hip_fatbin_<id>`, the
    # link step of a full build supplies it); here there is no device code, so an empty image stands in for each --
    # the runtime only looks inside an image at the first kernel launch, which this build never makes
    nm = subprocess.run(["nm", "-u"] + objs, capture_output=True, text=True, check=True).stdout
    ids = sorted({tok for tok in nm.split() if tok.startswith("__hip_fatbin_")})
    stub = os.path.join(obj_dir, "empty_images.c")
    with open(stub, "w") as f:
        f.write("/* generated by qiddm_amd.build.build_asan */\n")
        for sym in ids:
            f.write(f'const char {sym}[4096] __attribute__((aligned(4096), section(".hip_fatbin"))) = {{0}};\n')
    stub_o = stub[:-2] + ".o"
    subprocess.run(["gcc", "-c", "-fPIC", stub, "-o", stub_o], check=True)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-fsanitize=address", "-shared-libsan", "-fPIC", "-shared",
                    "-o", ASAN_LIB] + objs + [stub_o], check=True, cwd=CSRC)
    return ASAN_LIB


if __name__ == "__main__":
    if "--asan" in sys.argv:
        print(build_asan(verbose=True))
    else:
        build(force="--force" in sys.argv)
        print(LIB)
