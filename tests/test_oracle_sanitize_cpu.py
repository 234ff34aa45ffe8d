"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU ASan is not available)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    asan, ubsan = _lib("libasan.so"), _lib("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc sanitizer runtimes not installed")
    so = str(tmp_path / "libgeot_oracle_asan.so")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fopenmp",
                           "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-shared", "-o", so,
                           os.path.join(ROOT, "oracle", "geot_oracle.c"), "-lm"])
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "oracle_sanitize.py"), so], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "sanitizer run complete" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
