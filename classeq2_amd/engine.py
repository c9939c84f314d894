"""ctypes binding of libclsplace.so -- the C-ABI of include/cls_place.h.

Host-side plumbing only: every placement runs in the HIP kernels behind
`cls_place_batch*`; there is no CPU fallback, and a missing or unloadable
extension raises instead of degrading.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import _abi
from .flatdb import FlatDb

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libclsplace.so")
_LIB = None

EXPORTS = [
    "cls_device_count", "cls_db_create", "cls_db_validate", "cls_db_destroy", "cls_db_info_get", "cls_db_kernel_time", "cls_place_batch",
    "cls_place_batch_device", "cls_place_batch_stats", "cls_fasta_parse", "cls_fasta_free", "cls_last_error",
    "cls_version",
]


class ClsError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code = code
        self.msg = msg


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the placement path has no CPU fallback)"
            )
        L = C.CDLL(LIB_PATH)
        vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
        L.cls_device_count.restype = i32
        L.cls_db_create.argtypes = [C.POINTER(_abi.DbDesc), i32, C.POINTER(vp)]
        L.cls_db_create.restype = i32
        L.cls_db_validate.argtypes = [C.POINTER(_abi.DbDesc)]
        L.cls_db_validate.restype = i32
        L.cls_db_destroy.argtypes = [vp]
        L.cls_db_destroy.restype = None
        L.cls_db_info_get.argtypes = [vp, C.POINTER(_abi.DbInfo)]
        L.cls_db_info_get.restype = i32
        L.cls_db_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), i32]
        L.cls_db_kernel_time.restype = i32
        L.cls_place_batch.argtypes = [vp, vp, vp, u32, C.POINTER(_abi.Params), vp]
        L.cls_place_batch.restype = i32
        L.cls_place_batch_stats.argtypes = [vp, vp, vp, u32, C.POINTER(_abi.Params), vp, vp]
        L.cls_place_batch_stats.restype = i32
        L.cls_place_batch_device.argtypes = [vp, vp, vp, u32, C.POINTER(_abi.Params), vp, vp, vp]
        L.cls_place_batch_device.restype = i32
        L.cls_fasta_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(_abi.Fasta)]
        L.cls_fasta_parse.restype = i32
        L.cls_fasta_free.argtypes = [C.POINTER(_abi.Fasta)]
        L.cls_fasta_free.restype = None
        L.cls_last_error.restype = C.c_char_p
        L.cls_version.restype = C.c_char_p
        _LIB = L
    return _LIB


def _check(rc: int):
    if rc != 0:
        raise ClsError(rc, lib().cls_last_error().decode(errors="replace"))


def make_params(max_iterations: Optional[int] = None, min_match_coverage: Optional[float] = None,
                remove_intersection: Optional[bool] = None) -> _abi.Params:
    """The three Option<> arguments of place_sequence (place_sequence.rs:46-48)."""
    p = _abi.Params()
    if max_iterations is not None:
        p.flags |= _abi.HAS_MAX_ITERATIONS
        p.max_iterations = max_iterations
    if min_match_coverage is not None:
        p.flags |= _abi.HAS_MIN_MATCH_COVERAGE
        p.min_match_coverage = min_match_coverage
    if remove_intersection is not None:
        p.flags |= _abi.HAS_REMOVE_INTERSECTION
        p.remove_intersection = 1 if remove_intersection else 0
    return p


def device_count() -> int:
    return lib().cls_device_count()


def validate(flat: FlatDb) -> None:
    d = flat.desc()
    _check(lib().cls_db_validate(C.byref(d)))


def fasta_parse(text: bytes):
    """-> (headers: list[bytes], bases u8[], offsets u64[n+1], truncated: bool); a1 semantics."""
    f = _abi.Fasta()
    _check(lib().cls_fasta_parse(text, len(text), C.byref(f)))
    try:
        n = f.n
        hoff = np.ctypeslib.as_array(f.header_off, shape=(n + 1,)).copy()
        boff = np.ctypeslib.as_array(f.base_off, shape=(n + 1,)).copy()
        hraw = C.string_at(f.headers, int(hoff[-1]))
        bases = np.frombuffer(C.string_at(f.bases, int(boff[-1])), dtype=np.uint8).copy()
        headers = [hraw[int(hoff[i]) : int(hoff[i + 1])] for i in range(n)]
        return headers, bases, boff, bool(f.truncated)
    finally:
        lib().cls_fasta_free(C.byref(f))


class PlacementDb:
    """Owned handle on a device-resident index (cls_db)."""

    def __init__(self, flat: FlatDb, device: int = -1):
        self._h = C.c_void_p()
        d = flat.desc()
        _check(lib().cls_db_create(C.byref(d), device, C.byref(self._h)))
        info = _abi.DbInfo()
        _check(lib().cls_db_info_get(self._h, C.byref(info)))
        self.info = info

    def close(self):
        if getattr(self, "_h", None):
            lib().cls_db_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def place_batch(self, bases: np.ndarray, offsets: np.ndarray, params: Optional[_abi.Params] = None,
                    want_stats: bool = False):
        """Host buffers in, host records out (cls_place_batch / cls_place_batch_stats)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        out = np.zeros(n, dtype=_abi.PLACEMENT_DTYPE)
        pp = C.byref(params) if params is not None else None
        if want_stats:
            stats = np.zeros(n, dtype=_abi.STATS_DTYPE)
            _check(lib().cls_place_batch_stats(self._h, bases.ctypes.data, offsets.ctypes.data, n, pp,
                                               out.ctypes.data, stats.ctypes.data))
            return out, stats
        _check(lib().cls_place_batch(self._h, bases.ctypes.data, offsets.ctypes.data, n, pp, out.ctypes.data))
        return out

    def kernel_time(self, reset: bool = False):
        """(sum of ms, launches) of the dominant placement kernel since the last reset (cls_db_kernel_time)."""
        ms, cnt = C.c_double(0), C.c_uint64(0)
        _check(lib().cls_db_kernel_time(self._h, C.byref(ms), C.byref(cnt), 1 if reset else 0))
        return ms.value, cnt.value

    def place_batch_device(self, d_bases: int, d_offsets: int, n: int, d_out: int, params: Optional[_abi.Params] = None,
                           d_stats: int = 0, stream: int = 0) -> None:
        """Device pointers in/out, asynchronous on `stream` (cls_place_batch_device)."""
        pp = C.byref(params) if params is not None else None
        _check(lib().cls_place_batch_device(self._h, d_bases, d_offsets, n, pp, d_out, d_stats or None, stream or None))
