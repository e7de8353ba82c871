"""Device-resident vectors over the handle API of the C ABI (fpgpu_vec_*): HBM-resident operands
for chains of batch operations, and the multi-GPU layouts (sharded / replicated over the device
list of `set_devices`).  The reference keeps vectors in host memory (`Vec<T>`,
`DenseRowMatrix<T>::data`, src/matrix.rs:135-145); this is the batch extension's device-side
container.  Nothing here computes on the CPU."""
import ctypes

import numpy as np

from . import _lib
from ._lib import PLACE_CURRENT, PLACE_REPLICATED, PLACE_SHARDED, check  # noqa: F401


def set_devices(ids):
    """fpgpu_set_devices: the device list that sharded / replicated vectors are laid out over
    (ids may repeat; an empty list returns to "the current device alone")."""
    ids = list(ids)
    arr = (ctypes.c_int * max(1, len(ids)))(*ids)
    check(_lib.load().fpgpu_set_devices(arr, len(ids)))


def get_devices():
    lib = _lib.load()
    n = lib.fpgpu_get_devices(None, 0)
    arr = (ctypes.c_int * max(1, n))()
    lib.fpgpu_get_devices(arr, n)
    return list(arr[:n])


class DeviceSet:
    """An fpgpu_devset: a device list as a handle, for callers that must not depend on the process-wide
    default list of `set_devices` (pass it as `on=` to the allocators)."""

    def __init__(self, ids):
        self._lib = _lib.load()
        ids = list(ids)
        arr = (ctypes.c_int * max(1, len(ids)))(*ids)
        h = ctypes.c_void_p()
        check(self._lib.fpgpu_devset_create(arr, len(ids), ctypes.byref(h)))
        self._h = h

    def devices(self):
        n = self._lib.fpgpu_devset_devices(self._h, None, 0)
        arr = (ctypes.c_int * max(1, n))()
        self._lib.fpgpu_devset_devices(self._h, arr, n)
        return list(arr[:n])

    def free(self):
        if getattr(self, "_h", None):
            self._lib.fpgpu_devset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _set_handle(on):
    return on._h if on is not None else None


class DeviceVector:
    """An fpgpu_vec: `n` field elements of `field` in HBM (one device, sharded, or replicated)."""

    def __init__(self, field, handle):
        self.field, self._h, self._lib = field, handle, field._lib

    @classmethod
    def alloc(cls, field, n, placement=PLACE_CURRENT, on=None):
        h = ctypes.c_void_p()
        check(field._lib.fpgpu_vec_alloc_on(field._h, _set_handle(on), n, placement, ctypes.byref(h)))
        return cls(field, h)

    @classmethod
    def alloc_rows(cls, field, rows, cols, placement=PLACE_CURRENT, on=None):
        h = ctypes.c_void_p()
        check(field._lib.fpgpu_vec_alloc_rows_on(field._h, _set_handle(on), rows, cols, placement, ctypes.byref(h)))
        return cls(field, h)

    @classmethod
    def from_host(cls, field, a, placement=PLACE_CURRENT, row_len=None, on=None):
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, field.n64)
        v = cls.alloc(field, a.shape[0], placement, on) if not row_len else \
            cls.alloc_rows(field, a.shape[0] // row_len, row_len, placement, on)
        v.upload(a)
        return v

    def __len__(self):
        return int(self._lib.fpgpu_vec_len(self._h))

    def free(self):
        if getattr(self, "_h", None):
            self._lib.fpgpu_vec_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def parts(self):
        """[(device, first, count, device pointer, stream)] of every part."""
        out = []
        for i in range(self._lib.fpgpu_vec_parts(self._h)):
            dev, first, count = ctypes.c_int(), ctypes.c_size_t(), ctypes.c_size_t()
            ptr, stream = ctypes.c_void_p(), ctypes.c_void_p()
            check(self._lib.fpgpu_vec_part(self._h, i, ctypes.byref(dev), ctypes.byref(first), ctypes.byref(count),
                                           ctypes.byref(ptr), ctypes.byref(stream)))
            out.append((dev.value, first.value, count.value, ptr.value, stream.value))
        return out

    def upload(self, a, first=0):
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, self.field.n64)
        check(self._lib.fpgpu_vec_upload(self._h, ctypes.c_void_p(a.ctypes.data), first, a.shape[0]))
        return self

    def download(self, first=0, n=None):
        n = len(self) - first if n is None else n
        out = np.empty((n, self.field.n64), dtype=np.uint64)
        check(self._lib.fpgpu_vec_download(self._h, ctypes.c_void_p(out.ctypes.data), first, n))
        return out

    def sync(self):
        check(self._lib.fpgpu_vec_sync(self._h))
        return self

    def copy_from(self, src):
        check(self._lib.fpgpu_vec_copy(self._h, src._h))
        return self

    def fill_random(self, seed):
        check(self._lib.fpgpu_vec_fill_random(self.field._h, self._h, seed))
        return self


class DeviceIndices:
    """An fpgpu_idx: the 32-bit column indices of a SparseMatrix (rows x density), laid out like its values."""

    def __init__(self, rows, density, placement=PLACE_CURRENT, on=None):
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        check(self._lib.fpgpu_idx_alloc_rows_on(_set_handle(on), rows, density, placement, ctypes.byref(h)))
        self._h, self.n = h, rows * density

    def upload(self, col_idx, first=0):
        c = np.ascontiguousarray(col_idx, dtype=np.uint32).reshape(-1)
        check(self._lib.fpgpu_idx_upload(self._h, ctypes.c_void_p(c.ctypes.data), first, c.shape[0]))
        return self

    def free(self):
        if getattr(self, "_h", None):
            self._lib.fpgpu_idx_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
