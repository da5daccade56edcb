"""The CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (CPU side only: the GPU pool has no sanitizer runs).

The oracle's own tests (golden vectors, every search path, the builder, the stats fold) run once more in a child process against
an -fsanitize=address,undefined build of oracle/hnsw_oracle.c; any report aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_tests_pass_under_asan_and_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.exists(asan):
        pytest.skip("no libasan beside this gcc")
    env = dict(os.environ, ORACLE_SANITIZE="1", LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"), os.path.join(ROOT, "tests", "test_oracle_paths.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
