"""The oracle (plain C) under AddressSanitizer / UBSan: SURVEY 5's sanitizer row.  The sanitised library needs libasan preloaded
into the interpreter, so the run is a subprocess (tools/oracle_asan_run.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_clean_under_asan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this image")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"])
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "oracle_asan_run.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "asan ok" in r.stdout, r.stderr[-2000:]
