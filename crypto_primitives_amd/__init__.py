"""crypto_primitives_amd -- MI355X (gfx950) backend for crypto-primitives' prime-field
Montgomery hot path.  The product is `libfpgpu.so` (C ABI, include/fpgpu.h); this
package is its Python loader and a thin batch mirror of the reference's field
config interface.  (The directory is importable as `crypto_primitives_amd`; the
hyphenated name `crypto-primitives_amd` is not a valid Python identifier.)"""
from ._lib import FpgpuError, LIB_PATH, load  # noqa: F401
from .field import InvalidModulus, MontyField  # noqa: F401
from .vec import DeviceIndices, DeviceSet, DeviceVector, PLACE_CURRENT, PLACE_REPLICATED, PLACE_SHARDED, get_devices, set_devices  # noqa: F401

# Moduli named in BASELINE.json's configs.
BN254_FR = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
BLS12_381_FQ = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
GOLDILOCKS = 0xFFFFFFFF00000001
