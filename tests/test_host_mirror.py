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


STRESS = os.path.join(ROOT, "tests", "cpp", "stress_host_layers")


@pytest.mark.gpu
def test_host_layers_under_sixteen_threads():
    """tests/cpp/stress_host_layers.cpp: sixteen host threads of host-buffer calls, vectors sharded over device sets that
    name GPU 0 three and five times, partial transfers across part boundaries, gathered and row-sharded matrix products,
    sparse products and inversions on handles, device sets destroyed under live vectors -- every sharded result against
    the one-device call.  (tools/build_san.sh builds the same driver against an AddressSanitizer build of the host layers.)"""
    assert os.path.exists(STRESS), "tests/cpp/stress_host_layers must be prebuilt (__graft_entry__.build())"
    r = subprocess.run([STRESS, "16", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout
