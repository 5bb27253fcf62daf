"""The emulation suites once more against an AddressSanitizer + UBSan build of tests/emu (test tooling): host memory plays the LDS, the
padding behind every array of the kernels' bump allocator is poisoned (tests/emu/ptx_platform_emu.h), so an off-by-one of the kernel
logic — which the GPU would absorb silently in the next array's slack — is reported here."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ["tests/test_emu_parity.py", "tests/test_emu_partial_orders.py", "tests/test_emu_generate.py", "tests/test_emu_change.py", "tests/test_emu_patches.py", "tests/test_emu_rootmap.py", "tests/test_emu_biglog.py"]


def _lib(name):
    try:
        p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True, timeout=30).stdout.strip()
    except (OSError, subprocess.TimeoutExpired):
        return None
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_emulation_suites_under_asan_and_ubsan(tmp_path):
    asan, ubsan = _lib("libasan.so"), _lib("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("no libasan / libubsan in this image")
    lib = str(tmp_path / "libperitext_emu_asan.so")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-shared", "-fPIC", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined",
                    "-o", lib, os.path.join(ROOT, "tests", "emu", "emu_driver.cc")], check=True, timeout=600)
    env = dict(os.environ, PTX_EMU_LIB=lib, LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-s", "-p", "no:cacheprovider"] + SUITES, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "passed" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
