"""CPU sanitizer job (the reference has none; GPU sanitizers are not available on this pool): the oracle's full region
pipeline, the wire-format host decoder under a 20 000-case fuzz loop and the C++ tile plans, compiled with ASan + UBSan."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_host_logic_are_clean_under_asan_ubsan():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "oracle region under sanitizers: ok" in r.stdout and "no memory error" in r.stdout and "runtime error" not in r.stdout + r.stderr
