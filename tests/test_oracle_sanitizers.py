"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (host-side sanitizers;
GPU sanitizers are not available on this pool).  `make -C oracle sanitize` builds oracle/sph_oracle.c and a small
driver (oracle/oracle_selftest.c) with -fsanitize=address,undefined; the driver walks every entry point the parity
tests use over box-face, corner-cell, coincident and crowded inputs, clicks at the window's corners, both key orders."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None and shutil.which("cc") is None, reason="no C compiler")
def test_oracle_is_clean_under_asan_and_ubsan():
    b = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([os.path.join(ROOT, "oracle", "_san", "oracle_selftest")], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0 and "oracle self-test: done" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
