import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
FIELD_NAMES = ["bn254_fr", "bls12_381_fq", "goldilocks", "p64_ref_tests", "secp256k1_p", "bench_120bit",
               "odd_composite_128"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_cache = {}


def load_golden(name):
    if name not in _cache:
        with open(os.path.join(GOLDEN_DIR, name + ".json")) as f:
            _cache[name] = json.load(f)
    return _cache[name]


def h2i(s):
    return int(s, 16)


def ints_to_array(vals, n64):
    out = np.zeros((len(vals), n64), dtype=np.uint64)
    for i, v in enumerate(vals):
        for l in range(n64):
            out[i, l] = (v >> (64 * l)) & 0xFFFFFFFFFFFFFFFF
    return out


def array_to_ints(a):
    a = np.asarray(a, dtype=np.uint64)
    a = a.reshape(-1, a.shape[-1])
    return [sum(int(w) << (64 * l) for l, w in enumerate(row)) for row in a]


def splitmix64(z):
    M = 0xFFFFFFFFFFFFFFFF
    z = (z + 0x9E3779B97F4A7C15) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return z ^ (z >> 31)


def synth_elements(p, n64, n, seed):
    """Host twin of fpgpu_fill_random_dev (SURVEY.md section 8d generator)."""
    mask = (1 << p.bit_length()) - 1
    out = []
    for i in range(n):
        v = 0
        for l in range(n64):
            v |= splitmix64(seed ^ (i * n64 + l)) << (64 * l)
        v &= mask
        if v >= p:
            v -= p
        out.append(v)
    return out


@pytest.fixture(params=FIELD_NAMES)
def golden(request):
    return load_golden(request.param)


@pytest.fixture
def tuning():
    """Sets tuning / test hooks of the library (fpgpu_tuning_set) and restores the defaults afterwards.
    The library reads the FPGPU_* environment once, at first use, so tests go through this instead."""
    from crypto_primitives_amd import _lib
    touched = []

    def set_hook(name, value):
        touched.append(name)
        _lib.tuning_set(name, int(value))

    yield set_hook
    for name in touched:
        _lib.tuning_set(name)
