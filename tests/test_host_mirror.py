"""The C++ host-side mirror of the reference's field interface (crypto_primitives_amd/host/).
CPU part: scalar API + error behaviour, shaped like the reference's own tests.
GPU part: every batch method (which goes through the C ABI to the MI355X) against the
scalar methods."""
import os
import subprocess

import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "tests", "cpp", "test_monty_field")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)


def test_host_mirror_scalar_api():
    _build()
    r = subprocess.run([EXE, "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout


@pytest.mark.gpu
def test_host_mirror_batch_api_on_gpu():
    assert os.path.exists(EXE), "tests/cpp/test_monty_field must be prebuilt (__graft_entry__.build())"
    r = subprocess.run([EXE, "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout
