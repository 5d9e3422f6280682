"""AddressSanitizer + UBSan over the product's host-side C++ (CPU only; GPU sanitizers are
not available on the pool).  Builds tests/emu/sanitize_main.cpp with the product sources."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_code_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "sanitize_main")
    csrc = os.path.join(ROOT, "suchtree_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "emu", "sanitize_main.cpp"),
           os.path.join(csrc, "tree_prep.cpp"), os.path.join(csrc, "newick_parse.cpp")]
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "sanitize ok" in out.stdout


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_copy_pool_under_thread_sanitizer(tmp_path):
    """The host path's thread pool (spin-then-sleep dispatch, host_pipe.h) under ThreadSanitizer:
    hundreds of back-to-back phases, pauses that put the workers to sleep, stop and restart."""
    exe = str(tmp_path / "pool_main")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc, "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-Wno-option-ignored", "-o", exe,
                           os.path.join(ROOT, "tests", "emu", "pool_main.cpp"), "-lpthread"], cwd=str(tmp_path))
    out = subprocess.run([exe, "600"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "pool ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    assert "ThreadSanitizer" not in out.stderr
