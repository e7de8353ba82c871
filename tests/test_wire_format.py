"""The element wire format against the ABI buffers.

The reference's serde impls hand an element's Montgomery representative to crypto-bigint's own
(de)serialiser (MontyFieldElement: src/field/crypto_bigint_monty.rs:391-415; ConstMontyField:
src/field/crypto_bigint_const_monty.rs:662-690; BoxedMontyFieldElement:
src/field/crypto_bigint_boxed_monty.rs:448-470; Uint: src/semiring/crypto_bigint_uint.rs:668-695),
which writes the integer's little-endian bytes (`Encoding::to_le_bytes`).  A vector on the wire is
those encodings back to back.  The C ABI takes n64 little-endian u64 words per element, back to
back -- on this little-endian host the same bytes, so a received buffer can be handed to the
library as it lies and a result buffer can be sent as it lies.  Checked for the three adapters'
widths: MontyField n64 = 1 (Goldilocks), ConstMontyField n64 = 4 (BN254 Fr), BoxedMontyField
n64 = 6 (BLS12-381 Fq at 384-bit precision).

PARITY UNPINNED: the expectation below is a restatement (Python's int.to_bytes(..., "little")) of what
crypto-bigint's serialiser is understood to write; the reference holds no byte fixture of a serialised
element and crypto-bigint is not in its tree, so this is not a check against the reference itself."""
import sys

import numpy as np
import pytest

from conftest import array_to_ints, h2i, ints_to_array, load_golden

ADAPTERS = [("goldilocks", "MontyField<1>"), ("bn254_fr", "ConstMontyField<_, 4>"), ("bls12_381_fq", "BoxedMontyField, 384 bits")]


def wire_encode(values, n64):
    """crypto-bigint `to_le_bytes` of every element, concatenated."""
    return b"".join(int(v).to_bytes(8 * n64, "little") for v in values)


def wire_decode(buf, n64):
    size = 8 * n64
    assert len(buf) % size == 0
    return [int.from_bytes(buf[i:i + size], "little") for i in range(0, len(buf), size)]


@pytest.mark.parametrize("name,adapter", ADAPTERS)
def test_wire_bytes_are_the_abi_buffer(name, adapter):
    assert sys.byteorder == "little"
    g = load_golden(name)
    n64 = g["n64"]
    # Montgomery representatives out of the golden vectors (products, i.e. arbitrary canonical residues)
    vals = [h2i(v["mul"]) for v in g["binary"]] + [0, 1, h2i(g["modulus"]) - 1, h2i(g["one"]), h2i(g["r2"])]
    wire = wire_encode(vals, n64)
    # wire -> ABI buffer: reinterpret, no conversion
    abi = np.frombuffer(wire, dtype="<u8").reshape(-1, n64)
    assert np.array_equal(abi, ints_to_array(vals, n64))
    assert array_to_ints(abi) == vals
    # ABI buffer -> wire: the array's memory is the encoding
    assert ints_to_array(vals, n64).tobytes() == wire
    assert wire_decode(ints_to_array(vals, n64).tobytes(), n64) == vals


@pytest.mark.parametrize("name,adapter", ADAPTERS)
def test_context_parameters_come_back_in_wire_order(name, adapter):
    """fpgpu_ctx_params writes modulus / one / r2 into caller buffers in the same layout."""
    import crypto_primitives_amd as cpa
    g = load_golden(name)
    f = cpa.MontyField(h2i(g["modulus"]), g["n64"])
    assert f.one_words.tobytes() == wire_encode([h2i(g["one"])], g["n64"])
    assert f.r2_words.tobytes() == wire_encode([h2i(g["r2"])], g["n64"])
    assert f.modulus_words.tobytes() == wire_encode([h2i(g["modulus"])], g["n64"])


@pytest.mark.gpu
@pytest.mark.parametrize("name,adapter", ADAPTERS)
def test_wire_buffer_round_trips_through_the_device(name, adapter):
    """A received wire buffer goes to the device as it lies (host-buffer call and handle upload), and the
    result buffer is the wire encoding of the results: lift(project(x)) == x and a * one == a."""
    import crypto_primitives_amd as cpa
    g = load_golden(name)
    p, n64 = h2i(g["modulus"]), g["n64"]
    f = cpa.MontyField(p, n64)
    vals = [h2i(v["a"]) % p for v in g["binary"]][:200] + [0, 1, p - 1]
    wire = wire_encode(vals, n64)
    buf = np.frombuffer(wire, dtype="<u8").reshape(-1, n64)
    assert f.lift(f.project(buf)).tobytes() == wire
    one = np.tile(f.one_words, (len(vals), 1))
    assert f.mul(buf, one).tobytes() == wire
    v = cpa.DeviceVector.from_host(f, buf)
    assert v.download().tobytes() == wire
    want = wire_encode([(p - x) % p for x in vals], n64)
    out = cpa.DeviceVector.alloc(f, len(vals))
    f.neg_vec(v, out)
    assert out.download().tobytes() == want
