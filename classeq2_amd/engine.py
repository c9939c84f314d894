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
# (CLS_PLACE_LIB: an experiment build of the same library, tools/build_variant.sh)
LIB_PATH = os.environ.get("CLS_PLACE_LIB") or os.path.join(_HERE, "csrc", "libclsplace.so")
_LIB = None

EXPORTS = [
    "cls_device_count", "cls_db_create", "cls_db_validate", "cls_db_destroy", "cls_db_info_get", "cls_db_info_get2", "cls_db_kernel_time", "cls_db_kernel_name",
    "cls_db_set_max_read_len", "cls_place_batch",
    "cls_place_batch_device", "cls_place_batch_stats", "cls_fasta_parse", "cls_fasta_free", "cls_fasta_scan_device", "cls_fasta_dev_free",
    "cls_fasta_parse_gpu", "cls_place_fasta_text", "cls_last_error",
    "cls_version", "cls_set_tuning", "cls_tuning_from_env",
]
HOST_EXPORTS = [
    "cls_tree_load_json", "cls_tree_load", "cls_tree_init_from_file", "cls_tree_from_newick", "cls_tree_serialize", "cls_tree_save", "cls_tree_free", "cls_tree_set_annotations_yaml", "cls_tree_build_kmers_map", "cls_tree_desc", "cls_serialize_results",
    "cls_host_free", "cls_place_sequences", "cls_host_last_error",
]
SERVICE_EXPORTS = [
    "cls_service_create", "cls_service_destroy", "cls_service_add_model", "cls_service_submit", "cls_service_wait", "cls_service_pause",
    "cls_service_stats_get",
]
FORMAT_YAML, FORMAT_JSONL = 0, 1
DB_FORMAT_ZSTD, DB_FORMAT_YAML, DB_FORMAT_JSON = 0, 1, 2


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
        L.cls_db_kernel_name.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.cls_db_kernel_name.restype = i32
        L.cls_db_set_max_read_len.argtypes = [vp, C.c_uint64]
        L.cls_db_set_max_read_len.restype = i32
        L.cls_place_batch.argtypes = [vp, vp, vp, u32, C.POINTER(_abi.Params), vp]
        L.cls_place_batch.restype = i32
        L.cls_place_batch_stats.argtypes = [vp, vp, vp, u32, C.POINTER(_abi.Params), vp, vp]
        L.cls_place_batch_stats.restype = i32
        L.cls_place_batch_device.argtypes = [vp, vp, vp, u32, C.POINTER(_abi.Params), vp, vp, vp]
        L.cls_place_batch_device.restype = i32
        L.cls_fasta_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(_abi.Fasta)]
        L.cls_fasta_parse.restype = i32
        L.cls_fasta_parse_gpu.argtypes = [C.c_char_p, C.c_size_t, i32, C.POINTER(_abi.Fasta)]
        L.cls_fasta_parse_gpu.restype = i32
        L.cls_place_fasta_text.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(_abi.Params), C.POINTER(_abi.Fasta), C.POINTER(vp)]
        L.cls_place_fasta_text.restype = i32
        L.cls_fasta_free.argtypes = [C.POINTER(_abi.Fasta)]
        L.cls_fasta_free.restype = None
        L.cls_last_error.restype = C.c_char_p
        L.cls_version.restype = C.c_char_p
        L.cls_set_tuning.argtypes = [C.c_char_p, i32]
        L.cls_set_tuning.restype = i32
        L.cls_tuning_from_env.argtypes = []
        L.cls_tuning_from_env.restype = None
        # host-side mirror (include/cls_host.h)
        L.cls_tree_load_json.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.cls_tree_load_json.restype = i32
        L.cls_tree_load.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.cls_tree_load.restype = i32
        L.cls_tree_init_from_file.argtypes = [C.c_char_p, C.c_double, C.POINTER(vp)]
        L.cls_tree_init_from_file.restype = i32
        L.cls_tree_from_newick.argtypes = [C.c_char_p, C.c_char_p, C.c_double, C.POINTER(vp)]
        L.cls_tree_from_newick.restype = i32
        L.cls_tree_serialize.argtypes = [vp, i32, i32, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.cls_tree_serialize.restype = i32
        L.cls_tree_save.argtypes = [vp, C.c_char_p, i32, i32]
        L.cls_tree_save.restype = i32
        L.cls_tree_free.argtypes = [vp]
        L.cls_tree_free.restype = None
        L.cls_tree_set_annotations_yaml.argtypes = [vp, C.c_char_p]
        L.cls_tree_set_annotations_yaml.restype = i32
        L.cls_tree_build_kmers_map.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint64, u32]
        L.cls_tree_build_kmers_map.restype = i32
        L.cls_tree_desc.argtypes = [vp, C.POINTER(_abi.DbDesc)]
        L.cls_tree_desc.restype = i32
        L.cls_serialize_results.argtypes = [vp, C.c_char_p, vp, u32, vp, i32, C.POINTER(vp), C.POINTER(C.c_size_t),
                                            C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.cls_serialize_results.restype = i32
        L.cls_host_free.argtypes = [vp]
        L.cls_host_free.restype = None
        L.cls_place_sequences.argtypes = [vp, vp, C.c_char_p, C.c_char_p, C.POINTER(_abi.Params), i32, i32,
                                          C.POINTER(u32), C.POINTER(C.c_double)]
        L.cls_place_sequences.restype = i32
        L.cls_host_last_error.restype = C.c_char_p
        # resident batching service (include/cls_service.h)
        L.cls_service_create.argtypes = [C.POINTER(vp)]
        L.cls_service_create.restype = i32
        L.cls_service_destroy.argtypes = [vp]
        L.cls_service_destroy.restype = None
        L.cls_service_add_model.argtypes = [vp, C.c_char_p, vp]
        L.cls_service_add_model.restype = i32
        L.cls_service_submit.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(_abi.Params), C.POINTER(C.c_uint64)]
        L.cls_service_submit.restype = i32
        L.cls_service_wait.argtypes = [vp, C.c_uint64, C.POINTER(_abi.Fasta), C.POINTER(vp)]
        L.cls_service_wait.restype = i32
        L.cls_service_pause.argtypes = [vp, i32]
        L.cls_service_pause.restype = i32
        L.cls_service_stats_get.argtypes = [vp, C.POINTER(_abi.ServiceStats)]
        L.cls_service_stats_get.restype = i32
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


def set_tuning(name: str, value: int) -> None:
    """An experiment knob of the library (csrc/cls_tuning.h); none changes a result."""
    _check(lib().cls_set_tuning(name.encode(), int(value)))


def tuning_from_env() -> None:
    """Take every knob from its CLS_* environment variable (tools/ and A/B runs; the library never does on its own)."""
    lib().cls_tuning_from_env()


def device_count() -> int:
    return lib().cls_device_count()


def validate(flat: FlatDb) -> None:
    d = flat.desc()
    _check(lib().cls_db_validate(C.byref(d)))


def fasta_parse(text: bytes, device: Optional[int] = None):
    """-> (headers: list[bytes], bases u8[], offsets u64[n+1], truncated: bool); a1 semantics.
    `device`: run the stage's data-parallel passes on that GPU (cls_fasta_parse_gpu) instead of the host parser."""
    f = _abi.Fasta()
    if device is None:
        _check(lib().cls_fasta_parse(text, len(text), C.byref(f)))
    else:
        _check(lib().cls_fasta_parse_gpu(text, len(text), device, C.byref(f)))
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

    def refresh_info(self):
        """Re-read cls_db_info (scratch_slots and max_read_kmers change over a handle's life)."""
        _check(lib().cls_db_info_get(self._h, C.byref(self.info)))
        return self.info

    def kernel_name(self) -> str:
        """Template instance of the dominant placement kernel this handle launches (cls_db_kernel_name)."""
        buf = C.create_string_buffer(256)
        _check(lib().cls_db_kernel_name(self._h, buf, len(buf)))
        return buf.value.decode()

    def place_batch(self, bases: np.ndarray, offsets: np.ndarray, params: Optional[_abi.Params] = None,
                    want_stats: bool = False, out: Optional[np.ndarray] = None):
        """Host buffers in, host records out (cls_place_batch / cls_place_batch_stats).  `out`: caller-owned record
        array (e.g. in pinned memory) to fill instead of a fresh one."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        if out is None:
            out = np.zeros(n, dtype=_abi.PLACEMENT_DTYPE)
        assert out.dtype == _abi.PLACEMENT_DTYPE and len(out) >= n and out.flags.c_contiguous
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

    def place_fasta_text(self, text: bytes, params: Optional[_abi.Params] = None):
        """FASTA text -> (headers, records, truncated) with the FASTA stage and the placement on the device
        (cls_place_fasta_text): the reads never return to the host."""
        f = _abi.Fasta()
        recs = C.c_void_p()
        pp = C.byref(params) if params is not None else None
        _check(lib().cls_place_fasta_text(self._h, text, len(text), pp, C.byref(f), C.byref(recs)))
        try:
            n = f.n
            hoff = np.ctypeslib.as_array(f.header_off, shape=(n + 1,)).copy()
            hraw = C.string_at(f.headers, int(hoff[-1]))
            headers = [hraw[int(hoff[i]) : int(hoff[i + 1])] for i in range(n)]
            out = np.frombuffer(C.string_at(recs, n * 24), dtype=_abi.PLACEMENT_DTYPE).copy() if n else np.zeros(0, _abi.PLACEMENT_DTYPE)
            return headers, out, bool(f.truncated)
        finally:
            lib().cls_fasta_free(C.byref(f))
            lib().cls_host_free(recs)

    def set_max_read_len(self, n_bases: int) -> None:
        """Longest read place_batch_device() provisions for (cls_db_set_max_read_len)."""
        _check(lib().cls_db_set_max_read_len(self._h, n_bases))
        _check(lib().cls_db_info_get(self._h, C.byref(self.info)))

    def place_batch_device(self, d_bases: int, d_offsets: int, n: int, d_out: int, params: Optional[_abi.Params] = None,
                           d_stats: int = 0, stream: int = 0) -> None:
        """Device pointers in/out, asynchronous on `stream` (cls_place_batch_device)."""
        pp = C.byref(params) if params is not None else None
        _check(lib().cls_place_batch_device(self._h, d_bases, d_offsets, n, pp, d_out, d_stats or None, stream or None))


def _check_host(rc: int):
    if rc != 0:
        msg = lib().cls_host_last_error().decode(errors="replace") or lib().cls_last_error().decode(errors="replace")
        raise ClsError(rc, msg)


class Tree:
    """cls_tree: the reference's database / tree JSON export + optional annotations (include/cls_host.h)."""

    def __init__(self, path: Optional[str] = None, annotations_yaml: Optional[str] = None, *, _handle=None):
        """`path`: a database / tree file in any form load_database reads (.cls zstd YAML, YAML, JSON)."""
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
        elif path.endswith(".json"):
            _check_host(lib().cls_tree_load_json(path.encode(), C.byref(self._h)))
        else:
            _check_host(lib().cls_tree_load(path.encode(), C.byref(self._h)))
        if annotations_yaml:
            _check_host(lib().cls_tree_set_annotations_yaml(self._h, annotations_yaml.encode()))

    @classmethod
    def from_newick_file(cls, tree_path: str, min_branch_support: float = 70.0) -> "Tree":
        """Tree::init_from_file: Newick -> sanitized tree (cls_tree_init_from_file)."""
        h = C.c_void_p()
        _check_host(lib().cls_tree_init_from_file(tree_path.encode(), min_branch_support, C.byref(h)))
        return cls(_handle=h)

    @classmethod
    def from_newick(cls, text: str, name: Optional[str] = None, min_branch_support: float = 70.0) -> "Tree":
        h = C.c_void_p()
        _check_host(lib().cls_tree_from_newick(text.encode(), name.encode() if name else None, min_branch_support, C.byref(h)))
        return cls(_handle=h)

    def dumps(self, fmt: int = DB_FORMAT_YAML, only_tree: bool = False) -> bytes:
        """`cls convert database` serialisation (cls_tree_serialize)."""
        buf, n = C.c_void_p(), C.c_size_t()
        _check_host(lib().cls_tree_serialize(self._h, fmt, 1 if only_tree else 0, C.byref(buf), C.byref(n)))
        try:
            return C.string_at(buf, n.value)
        finally:
            lib().cls_host_free(buf)

    def save(self, path: str, fmt: int = DB_FORMAT_ZSTD, only_tree: bool = False) -> None:
        _check_host(lib().cls_tree_save(self._h, path.encode(), fmt, 1 if only_tree else 0))

    def close(self):
        if getattr(self, "_h", None):
            lib().cls_tree_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build_kmers_map(self, msa_text: bytes, k_size: int, m_size: int = 4, reference_header_shift: bool = True,
                        forward_only: bool = False) -> None:
        """`cls build-db` (map_kmers_to_tree) on this tree; see include/cls_host.h."""
        flags = (1 if reference_header_shift else 0) | (2 if forward_only else 0)
        _check_host(lib().cls_tree_build_kmers_map(self._h, msa_text, len(msa_text), k_size, m_size, flags))

    def flat(self) -> FlatDb:
        d = _abi.DbDesc()
        _check_host(lib().cls_tree_desc(self._h, C.byref(d)))
        return FlatDb.from_desc(d, keepalive=self)

    def serialize(self, headers, records: np.ndarray, fmt: int = FORMAT_YAML):
        """-> (result text, error text) exactly as `place_sequences` would append them to its two files."""
        hb = [h if isinstance(h, bytes) else h.encode() for h in headers]
        off = np.concatenate([[0], np.cumsum([len(h) for h in hb])]).astype(np.uint64)
        recs = np.ascontiguousarray(records, dtype=_abi.PLACEMENT_DTYPE)
        out, err = C.c_void_p(), C.c_void_p()
        ol, el = C.c_size_t(0), C.c_size_t(0)
        _check_host(lib().cls_serialize_results(self._h, b"".join(hb), off.ctypes.data, len(hb), recs.ctypes.data, fmt,
                                                C.byref(out), C.byref(ol), C.byref(err), C.byref(el)))
        try:
            return C.string_at(out, ol.value), C.string_at(err, el.value)
        finally:
            lib().cls_host_free(out)
            lib().cls_host_free(err)


def place_sequences(db: "PlacementDb", tree: Tree, query_path: str, out_file: str, params: Optional[_abi.Params] = None,
                    overwrite: bool = False, fmt: int = FORMAT_YAML):
    """The whole use-case (mod.rs:43-270) through cls_place_sequences: -> (records read, seconds)."""
    n, sec = C.c_uint32(0), C.c_double(0)
    _check_host(lib().cls_place_sequences(db._h, tree._h, query_path.encode(), out_file.encode(),
                                          C.byref(params) if params is not None else None, 1 if overwrite else 0, fmt,
                                          C.byref(n), C.byref(sec)))
    return n.value, sec.value


class Service:
    """Resident batching service (include/cls_service.h): models stay on the device, waiting jobs share batches."""

    def __init__(self):
        self._h = C.c_void_p()
        _check(lib().cls_service_create(C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            lib().cls_service_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_model(self, model_id: str, db: "PlacementDb") -> None:
        """The service takes the handle over (`db` must not be used or closed afterwards)."""
        _check(lib().cls_service_add_model(self._h, model_id.encode(), db._h))
        db._h = C.c_void_p()

    def submit(self, model_id: str, fasta_text: bytes, params: Optional[_abi.Params] = None) -> int:
        t = C.c_uint64(0)
        _check(lib().cls_service_submit(self._h, model_id.encode(), fasta_text, len(fasta_text),
                                        C.byref(params) if params is not None else None, C.byref(t)))
        return t.value

    def wait(self, ticket: int):
        """-> (headers, records, truncated) of the job."""
        f = _abi.Fasta()
        recs = C.c_void_p()
        _check(lib().cls_service_wait(self._h, ticket, C.byref(f), C.byref(recs)))
        try:
            n = f.n
            hoff = np.ctypeslib.as_array(f.header_off, shape=(n + 1,)).copy()
            hraw = C.string_at(f.headers, int(hoff[-1]))
            headers = [hraw[int(hoff[i]) : int(hoff[i + 1])] for i in range(n)]
            out = np.frombuffer(C.string_at(recs, n * 24), dtype=_abi.PLACEMENT_DTYPE).copy() if n else np.zeros(0, _abi.PLACEMENT_DTYPE)
            return headers, out, bool(f.truncated)
        finally:
            lib().cls_fasta_free(C.byref(f))
            lib().cls_host_free(recs)

    def pause(self, paused: bool = True) -> None:
        _check(lib().cls_service_pause(self._h, 1 if paused else 0))

    def stats(self) -> dict:
        st = _abi.ServiceStats()
        _check(lib().cls_service_stats_get(self._h, C.byref(st)))
        return {k: int(getattr(st, k)) for k, _ in _abi.ServiceStats._fields_}
