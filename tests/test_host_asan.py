"""CPU: the host side of the C ABI under AddressSanitizer ("ASan host build + re-run").  `make asan` rebuilds the
three host translation units with -fsanitize=address (device code untouched: GPU sanitizers are unavailable on this pool) and
tests/host_abi_driver.c -- plain C against include/eeg2video_hip.h -- walks the host-only context through it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "eeg2video_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang"


@pytest.mark.skipif(not (os.path.isfile(CLANG) and shutil.which("make")), reason="ROCm clang not present")
def test_host_abi_under_address_sanitizer(tmp_path):
    subprocess.run(["make", "-C", CSRC, "-j", "8"], check=True, capture_output=True)          # kernel objects of the normal build
    r = subprocess.run(["make", "-C", CSRC, "asan", "-j", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    exe = str(tmp_path / "host_abi_driver")
    lib_dir = os.path.join(ROOT, "eeg2video_amd", "lib")
    r = subprocess.run([CLANG, "-std=c11", "-g", "-fsanitize=address", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "host_abi_driver.c"), "-o", exe, "-L", lib_dir, "-leeg2video_hip_asan",
                        "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", LSAN_OPTIONS="suppressions=" + os.path.join(ROOT, "tests", "lsan.supp"))
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "host_abi_driver: ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    assert "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
