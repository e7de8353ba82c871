"""include/fpgpu.h from plain C99 (tests/c/abi_c99.c, -std=c99 -pedantic): what cgo, bindgen or any other FFI
generator would see.  CPU part: contexts, parameters against the constants of SURVEY.md 8c, error codes, no CPU
fallback.  GPU part: host-buffer calls and device-vector handles checked with field laws."""
import os
import subprocess

import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "tests", "c", "abi_c99")


def test_header_is_c99_and_context_behaviour():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "c")], stdout=subprocess.DEVNULL)
    r = subprocess.run([EXE, "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "0 failures" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_c99_caller_on_gpu():
    assert os.path.exists(EXE), "tests/c/abi_c99 must be prebuilt (__graft_entry__.build())"
    r = subprocess.run([EXE, "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "0 failures" in r.stdout, r.stdout + r.stderr
