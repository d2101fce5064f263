"""The host side of the C ABI under AddressSanitizer (CPU build box only -- GPU ASan / xnack+ runs are not available on
the pool): `qiddm_amd.build.build_asan()` compiles the entry points, their argument validation and launch wrappers with
`--cuda-host-only -fsanitize=address`, and the invalid-descriptor / NULL-pointer / bad-size cases run against it in a
python that has the sanitizer runtime preloaded."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_invalid_arguments_are_rejected_cleanly_under_asan():
    import torch
    if torch.cuda.is_available():
        pytest.skip("sanitizer build is a CPU-box check")
    from qiddm_amd import build
    lib = build.build_asan()
    env = dict(os.environ)
    env.update({"LD_PRELOAD": build.asan_runtime(), "QIDDM_HIP_LIB": lib,
                "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=66",
                "PYTHONPATH": ROOT + os.pathsep + env.get("PYTHONPATH", "")})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_asan_invalid_args.py")], capture_output=True,
                       text=True, env=env, timeout=600, cwd=ROOT)
    assert "AddressSanitizer" not in p.stderr, p.stderr[-4000:]
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    assert "invalid calls rejected cleanly" in p.stdout
