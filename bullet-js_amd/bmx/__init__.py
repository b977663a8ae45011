"""bmx — ctypes binding of libbmx.so (include/bmx.h), the MI355X CRDT-merge / index-scan engine.

This is the Python-side driver used by tests and bench.py. The product surface is the C ABI and the
N-API/JS host (bullet-js_amd/js); nothing here computes anything on the CPU: every call goes to the
HIP library and raises BmxError if it (or a GPU) is missing.
"""
import ctypes as C
import os
import sys

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("BMX_LIB_PATH") or os.path.join(_PKG, "libbmx.so")   # the override exists for A/B measurements of two builds

OK = 0
ERR_INVALID, ERR_HIP, ERR_FULL, ERR_NOMEM, ERR_RANGE, ERR_INTERNAL, ERR_NO_DEVICE, ERR_NO_INDEX, ERR_OVERFLOW = -1, -2, -3, -4, -5, -6, -7, -8, -9
MEM_HOST, MEM_DEVICE = 0, 1
INSERT_REFERENCE, INSERT_DELTA = 0, 1
VAL_DELETED = -(1 << 63)   # BMX_VAL_DELETED: tombstone value of bmx_put_rows
MERGE_MARK_CREATED, APPLIED_CREATED, APPLIED_INDEX = 0x1000, 0x80000000, 0x00FFFFFF
MERGE_UNIQUE_KEYS = 0x100
MERGE_STRICT_FLAGS = 0x200
CTX_FIXED_CAPACITY = 2
FLAG_INCOMING, FLAG_CURRENT, FLAG_HISTORICAL = 1, 2, 4
MAX_BATCH = 1 << 24

EXPORTS = [
    "bmx_create", "bmx_create_ex", "bmx_destroy", "bmx_last_error", "bmx_abi_version", "bmx_selfcheck", "bmx_set_deferred_compaction", "bmx_set_side_stream", "bmx_set_wait_limit", "bmx_merge_fence", "bmx_get_deferred_counts", "bmx_get_info", "bmx_get_placement", "bmx_set_probe_waves", "bmx_sync", "bmx_set_stream", "bmx_get_stream", "bmx_seq_signal", "bmx_seq_wait",
    "bmx_load_rows", "bmx_put_rows", "bmx_merge_batch", "bmx_merge_submit", "bmx_merge_collect", "bmx_host_alloc", "bmx_host_free", "bmx_merge_records", "bmx_get_rows", "bmx_get_row", "bmx_dump_rows", "bmx_row_count", "bmx_reserve",
    "bmx_index_build", "bmx_index_drop", "bmx_index_size", "bmx_index_refresh_counts", "bmx_index_set_ordered", "bmx_index_ordered_info", "bmx_index_ordered_stats", "bmx_scan_range", "bmx_scan_equals", "bmx_scan_count", "bmx_scan_filter", "bmx_scan_range_pos", "bmx_index_ids",
    "bmx_owner_of", "bmx_partition_by_owner", "bmx_partition_by_owner_slabs", "bmx_partition_scatter", "bmx_merge_records_after", "bmx_ipc_alloc", "bmx_ipc_open", "bmx_ipc_close", "bmx_ipc_free", "bmx_seq_wait_all", "bmx_merge_tail_wait", "bmx_merge_notify", "bmx_timer_start", "bmx_timer_stop", "bmx_timer_mark", "bmx_timer_elapsed", "bmx_profile_enable", "bmx_profile_read", "bmx_profile_read_scan",
    "bmx_comm_create", "bmx_comm_destroy", "bmx_comm_last_error", "bmx_comm_nshards", "bmx_comm_shard", "bmx_comm_sync", "bmx_comm_load_rows", "bmx_comm_put_rows", "bmx_comm_merge",
    "bmx_comm_merge_dev", "bmx_comm_shard_result", "bmx_comm_row_count", "bmx_comm_get_rows", "bmx_comm_dump_rows", "bmx_comm_index_build", "bmx_comm_index_set_ordered",
    "bmx_comm_scan_range", "bmx_comm_scan_equals", "bmx_comm_scan_count", "bmx_comm_scan_filter",
    "bmx_vc_create", "bmx_vc_destroy", "bmx_vc_last_error", "bmx_vc_load_rows", "bmx_vc_merge_batch", "bmx_vc_get_rows", "bmx_vc_row_count", "bmx_vc_scan_range", "bmx_vc_merge_batch_dev", "bmx_vc_set_stream", "bmx_vc_sync",
    "bmx_vc_load_rows_ks", "bmx_vc_merge_batch_ks", "bmx_vc_get_rows_ks", "bmx_vc_merge_batch_ks_dev", "bmx_vc_keyset", "bmx_vc_keyset_dense",
]


class BmxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bmx error %d: %s" % (code, msg))
        self.code = code


class MergeStats(C.Structure):
    _fields_ = [("n_applied", C.c_uint64), ("n_conflicts", C.c_uint64), ("n_rows", C.c_uint64), ("reserved", C.c_uint64)]


class Term(C.Structure):
    _fields_ = [("field", C.c_uint32), ("reserved", C.c_uint32), ("lo", C.c_int64), ("hi", C.c_int64)]


class Info(C.Structure):
    _fields_ = [("capacity_rows", C.c_uint64), ("n_slots", C.c_uint64), ("table_bytes", C.c_uint64), ("n_rows", C.c_uint64),
                ("device", C.c_uint32), ("abi_version", C.c_uint32), ("n_indexes", C.c_uint32), ("epoch", C.c_uint32)]


DELTA_REC_DTYPE = np.dtype([("id", "<u8"), ("field", "<u4"), ("aux", "<u4"), ("ts", "<i8"), ("val", "<i8")])

# OR-ed into the flags of every Engine this process creates
DEFAULT_CTX_FLAGS = int(os.environ.get("BMX_CTX_FLAGS", "0"), 0)

_lib = None


def load_library():
    """dlopen libbmx.so and declare its signatures. No GPU is touched."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BmxError(ERR_NO_DEVICE, "libbmx.so not built (run __graft_entry__.build() or make -C bullet-js_amd): the engine has no CPU fallback")
    # Load order matters in a process that also uses PyTorch: its wheel bundles its own ROCm runtime (same sonames as /opt/rocm's). If
    # libbmx.so comes first it binds the system runtime and torch later brings a second one ("No HIP GPUs are available"); if torch comes
    # first, libbmx.so resolves libamdhip64.so.7 to the copy that is already loaded and both share streams and pointers.
    if "torch" not in sys.modules:
        try:
            import importlib.util
            if importlib.util.find_spec("torch") is not None:
                import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, i64, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int64, C.c_int
    L.bmx_create.argtypes = [i32, u64, u32, C.POINTER(vp)]; L.bmx_create.restype = i32
    L.bmx_create_ex.argtypes = [i32, u64, u32, u32, C.POINTER(vp)]; L.bmx_create_ex.restype = i32
    L.bmx_destroy.argtypes = [vp]; L.bmx_destroy.restype = None
    L.bmx_last_error.argtypes = [vp]; L.bmx_last_error.restype = C.c_char_p
    L.bmx_abi_version.argtypes = []; L.bmx_abi_version.restype = i32
    L.bmx_selfcheck.argtypes = [i32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]; L.bmx_selfcheck.restype = i32
    L.bmx_set_deferred_compaction.argtypes = [vp, i32]; L.bmx_set_deferred_compaction.restype = i32
    L.bmx_set_side_stream.argtypes = [vp, vp]; L.bmx_set_side_stream.restype = i32
    L.bmx_set_wait_limit.argtypes = [vp, C.c_double]; L.bmx_set_wait_limit.restype = i32
    L.bmx_merge_fence.argtypes = [vp]; L.bmx_merge_fence.restype = i32
    L.bmx_get_deferred_counts.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]; L.bmx_get_deferred_counts.restype = i32
    L.bmx_get_info.argtypes = [vp, C.POINTER(Info)]; L.bmx_get_info.restype = i32
    L.bmx_set_probe_waves.argtypes = [vp, i32]; L.bmx_set_probe_waves.restype = i32
    L.bmx_get_placement.argtypes = [vp, C.POINTER(u32), C.POINTER(C.c_float), C.POINTER(C.c_float)]; L.bmx_get_placement.restype = i32
    L.bmx_sync.argtypes = [vp]; L.bmx_sync.restype = i32
    L.bmx_set_stream.argtypes = [vp, vp]; L.bmx_set_stream.restype = i32
    L.bmx_get_stream.argtypes = [vp]; L.bmx_get_stream.restype = vp
    L.bmx_seq_signal.argtypes = [vp, vp, vp, u64]; L.bmx_seq_signal.restype = i32
    L.bmx_seq_wait.argtypes = [vp, vp, vp, u64]; L.bmx_seq_wait.restype = i32
    L.bmx_load_rows.argtypes = [vp, u64, vp, vp, vp, vp, i32]; L.bmx_load_rows.restype = i32
    L.bmx_put_rows.argtypes = [vp, u64, vp, vp, vp, vp, i32]; L.bmx_put_rows.restype = i32
    L.bmx_merge_batch.argtypes = [vp, u64, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]; L.bmx_merge_batch.restype = i32
    L.bmx_merge_records.argtypes = [vp, u64, vp, i32, vp, vp, vp, vp]; L.bmx_merge_records.restype = i32
    L.bmx_merge_submit.argtypes = [vp, u64, vp, vp, vp, vp, i32, i32, C.POINTER(u64)]; L.bmx_merge_submit.restype = i32
    L.bmx_merge_collect.argtypes = [vp, u64, vp, vp, vp, vp]; L.bmx_merge_collect.restype = i32
    L.bmx_get_rows.argtypes = [vp, u64, vp, vp, vp, vp, vp, i32]; L.bmx_get_rows.restype = i32
    L.bmx_get_row.argtypes = [vp, u64, u32, C.POINTER(i64), C.POINTER(i64)]; L.bmx_get_row.restype = i32
    L.bmx_dump_rows.argtypes = [vp, u64, vp, vp, vp, vp, vp, i32]; L.bmx_dump_rows.restype = i32
    L.bmx_row_count.argtypes = [vp, C.POINTER(u64)]; L.bmx_row_count.restype = i32
    L.bmx_reserve.argtypes = [vp, u64]; L.bmx_reserve.restype = i32
    L.bmx_index_build.argtypes = [vp, u32]; L.bmx_index_build.restype = i32
    L.bmx_index_drop.argtypes = [vp, u32]; L.bmx_index_drop.restype = i32
    L.bmx_index_size.argtypes = [vp, u32, C.POINTER(u64)]; L.bmx_index_size.restype = i32
    L.bmx_index_refresh_counts.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]; L.bmx_index_refresh_counts.restype = i32
    L.bmx_index_set_ordered.argtypes = [vp, u32, u32]; L.bmx_index_set_ordered.restype = i32
    L.bmx_index_ordered_info.argtypes = [vp, u32, C.POINTER(u32), C.POINTER(i32), C.POINTER(u64)]; L.bmx_index_ordered_info.restype = i32
    L.bmx_index_ordered_stats.argtypes = [vp, u32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(u64), C.POINTER(u64)]; L.bmx_index_ordered_stats.restype = i32
    L.bmx_scan_range.argtypes = [vp, u32, i64, i64, vp, u64, vp, i32]; L.bmx_scan_range.restype = i32
    L.bmx_scan_equals.argtypes = [vp, u32, i64, vp, u64, vp, i32]; L.bmx_scan_equals.restype = i32
    L.bmx_scan_count.argtypes = [vp, u32, i64, i64, vp, i32]; L.bmx_scan_count.restype = i32
    L.bmx_scan_filter.argtypes = [vp, u32, C.POINTER(Term), vp, u64, vp, i32]; L.bmx_scan_filter.restype = i32
    L.bmx_scan_range_pos.argtypes = [vp, u32, i64, i64, vp, u64, vp, i32]; L.bmx_scan_range_pos.restype = i32
    L.bmx_index_ids.argtypes = [vp, u32, u64, u64, vp, i32]; L.bmx_index_ids.restype = i32
    L.bmx_owner_of.argtypes = [u64, u32]; L.bmx_owner_of.restype = u32
    L.bmx_partition_by_owner.argtypes = [vp, u64, vp, vp, vp, vp, u32, vp, vp]; L.bmx_partition_by_owner.restype = i32
    L.bmx_partition_by_owner_slabs.argtypes = [vp, u64, vp, vp, vp, vp, u32, u64, vp, vp]; L.bmx_partition_by_owner_slabs.restype = i32
    L.bmx_partition_scatter.argtypes = [vp, u64, vp, vp, vp, vp, u32, u64, vp, vp, vp, u64, vp, u32, u64]; L.bmx_partition_scatter.restype = i32
    L.bmx_merge_records_after.argtypes = [vp, vp, u32, u64, u64, vp, i32, vp, vp, vp, vp]; L.bmx_merge_records_after.restype = i32
    L.bmx_ipc_alloc.argtypes = [vp, u64, u32, C.POINTER(vp), C.c_char_p]; L.bmx_ipc_alloc.restype = i32
    L.bmx_ipc_open.argtypes = [vp, C.c_char_p, i32, C.POINTER(vp)]; L.bmx_ipc_open.restype = i32
    L.bmx_ipc_close.argtypes = [vp, vp]; L.bmx_ipc_close.restype = i32
    L.bmx_ipc_free.argtypes = [vp, vp]; L.bmx_ipc_free.restype = i32
    L.bmx_host_alloc.argtypes = [C.c_uint64, C.POINTER(C.c_void_p)]; L.bmx_host_alloc.restype = i32
    L.bmx_host_free.argtypes = [vp]; L.bmx_host_free.restype = i32
    L.bmx_seq_wait_all.argtypes = [vp, vp, vp, u32, u64]; L.bmx_seq_wait_all.restype = i32
    L.bmx_merge_notify.argtypes = [vp, vp, u32]; L.bmx_merge_notify.restype = i32
    L.bmx_merge_tail_wait.argtypes = [vp, vp, u32, u64]; L.bmx_merge_tail_wait.restype = i32
    L.bmx_timer_start.argtypes = [vp]; L.bmx_timer_start.restype = i32
    L.bmx_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]; L.bmx_timer_stop.restype = i32
    L.bmx_timer_mark.argtypes = [vp]; L.bmx_timer_mark.restype = i32
    L.bmx_timer_elapsed.argtypes = [vp, C.POINTER(C.c_float)]; L.bmx_timer_elapsed.restype = i32
    L.bmx_profile_enable.argtypes = [vp, i32]; L.bmx_profile_enable.restype = i32
    L.bmx_profile_read.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(u32)]; L.bmx_profile_read.restype = i32
    L.bmx_profile_read_scan.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(u32)]; L.bmx_profile_read_scan.restype = i32
    L.bmx_comm_create.argtypes = [u32, C.POINTER(C.c_int), u64, u32, C.POINTER(vp)]; L.bmx_comm_create.restype = i32
    L.bmx_comm_destroy.argtypes = [vp]; L.bmx_comm_destroy.restype = None
    L.bmx_comm_last_error.argtypes = [vp]; L.bmx_comm_last_error.restype = C.c_char_p
    L.bmx_comm_nshards.argtypes = [vp]; L.bmx_comm_nshards.restype = u32
    L.bmx_comm_shard.argtypes = [vp, u32]; L.bmx_comm_shard.restype = vp
    L.bmx_comm_sync.argtypes = [vp]; L.bmx_comm_sync.restype = i32
    L.bmx_comm_load_rows.argtypes = [vp, u64, vp, vp, vp, vp]; L.bmx_comm_load_rows.restype = i32
    L.bmx_comm_put_rows.argtypes = [vp, u64, vp, vp, vp, vp]; L.bmx_comm_put_rows.restype = i32
    L.bmx_comm_merge.argtypes = [vp, u64, vp, vp, vp, vp, i32, vp, vp, vp]; L.bmx_comm_merge.restype = i32
    L.bmx_comm_merge_dev.argtypes = [vp, vp, vp, vp, vp, vp, i32, u64]; L.bmx_comm_merge_dev.restype = i32
    L.bmx_comm_shard_result.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]; L.bmx_comm_shard_result.restype = i32
    L.bmx_comm_row_count.argtypes = [vp, C.POINTER(u64)]; L.bmx_comm_row_count.restype = i32
    L.bmx_comm_get_rows.argtypes = [vp, u64, vp, vp, vp, vp, vp]; L.bmx_comm_get_rows.restype = i32
    L.bmx_comm_dump_rows.argtypes = [vp, u64, vp, vp, vp, vp, C.POINTER(u64)]; L.bmx_comm_dump_rows.restype = i32
    L.bmx_comm_index_build.argtypes = [vp, u32]; L.bmx_comm_index_build.restype = i32
    L.bmx_comm_index_set_ordered.argtypes = [vp, u32, u32]; L.bmx_comm_index_set_ordered.restype = i32
    L.bmx_comm_scan_range.argtypes = [vp, u32, i64, i64, vp, u64, C.POINTER(u64)]; L.bmx_comm_scan_range.restype = i32
    L.bmx_comm_scan_equals.argtypes = [vp, u32, i64, vp, u64, C.POINTER(u64)]; L.bmx_comm_scan_equals.restype = i32
    L.bmx_comm_scan_count.argtypes = [vp, u32, i64, i64, C.POINTER(u64)]; L.bmx_comm_scan_count.restype = i32
    L.bmx_comm_scan_filter.argtypes = [vp, u32, C.POINTER(Term), vp, u64, C.POINTER(u64)]; L.bmx_comm_scan_filter.restype = i32
    L.bmx_vc_create.argtypes = [i32, u64, u32, u32, C.POINTER(vp)]; L.bmx_vc_create.restype = i32
    L.bmx_vc_destroy.argtypes = [vp]; L.bmx_vc_destroy.restype = None
    L.bmx_vc_last_error.argtypes = [vp]; L.bmx_vc_last_error.restype = C.c_char_p
    L.bmx_vc_load_rows.argtypes = [vp, u64, vp, vp, vp, vp]; L.bmx_vc_load_rows.restype = i32
    L.bmx_vc_merge_batch.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp, vp]; L.bmx_vc_merge_batch.restype = i32
    L.bmx_vc_get_rows.argtypes = [vp, u64, vp, vp, vp, vp, vp]; L.bmx_vc_get_rows.restype = i32
    L.bmx_vc_row_count.argtypes = [vp, C.POINTER(u64)]; L.bmx_vc_row_count.restype = i32
    L.bmx_vc_scan_range.argtypes = [vp, u32, i64, i64, vp, u64, C.POINTER(u64)]; L.bmx_vc_scan_range.restype = i32
    L.bmx_vc_merge_batch_dev.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp, vp]; L.bmx_vc_merge_batch_dev.restype = i32
    L.bmx_vc_load_rows_ks.argtypes = [vp, u64, vp, vp, vp, vp, vp]; L.bmx_vc_load_rows_ks.restype = i32
    L.bmx_vc_merge_batch_ks.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp, vp, vp]; L.bmx_vc_merge_batch_ks.restype = i32
    L.bmx_vc_get_rows_ks.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp]; L.bmx_vc_get_rows_ks.restype = i32
    L.bmx_vc_merge_batch_ks_dev.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp, vp, vp]; L.bmx_vc_merge_batch_ks_dev.restype = i32
    L.bmx_vc_keyset.argtypes = [vp, C.c_uint32]; L.bmx_vc_keyset.restype = C.c_uint32
    L.bmx_vc_keyset_dense.argtypes = [C.c_uint32]; L.bmx_vc_keyset_dense.restype = C.c_uint32
    L.bmx_vc_set_stream.argtypes = [vp, vp]; L.bmx_vc_set_stream.restype = i32
    L.bmx_vc_sync.argtypes = [vp]; L.bmx_vc_sync.restype = i32
    _lib = L
    return L


def selfcheck(device=0):
    """bmx_selfcheck: (checked 16-byte loads, torn pairs seen, torn pairs of the split-store control). Raises if a pair tore."""
    L = load_library()
    r, t, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = L.bmx_selfcheck(int(device), C.byref(r), C.byref(t), C.byref(c))
    if rc:
        raise BmxError(rc, (L.bmx_last_error(None) or b"").decode())
    return r.value, t.value, c.value


def _np(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _ptr(a):
    """numpy array -> void*, torch tensor -> data_ptr, int passes through, None -> NULL."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    return C.c_void_p(a.data_ptr())


class HostBuffer:
    """Page-locked host memory (bmx_host_alloc) seen as a numpy array: inputs and outputs of host-array calls placed here move at the
    link's rate. Freed by close() or when collected; arrays taken from it must not outlive it."""

    def __init__(self, nbytes):
        self.L = load_library()
        p = C.c_void_p()
        rc = self.L.bmx_host_alloc(int(nbytes), C.byref(p))
        if rc:
            raise BmxError(rc, (self.L.bmx_last_error(None) or b"").decode())
        self.ptr, self.nbytes = p.value, int(nbytes)
        self._raw = (C.c_uint8 * self.nbytes).from_address(self.ptr)

    def array(self, dtype, count, offset=0):
        dt = np.dtype(dtype)
        if offset % dt.itemsize or offset + count * dt.itemsize > self.nbytes:
            raise ValueError("HostBuffer.array: out of range or misaligned")
        return np.frombuffer(self._raw, dtype=dt, count=count, offset=offset)

    def close(self):
        if self.ptr:
            self._raw = None
            self.L.bmx_host_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_columns(n):
    """-> (HostBuffer, id u64[n], field u32[n], ts i64[n], val i64[n]) in page-locked memory, the layout merge_batch() takes."""
    hb = HostBuffer(max(n, 1) * 28)
    return hb, hb.array(np.uint64, n), hb.array(np.uint32, n, 24 * n), hb.array(np.int64, n, 8 * n), hb.array(np.int64, n, 16 * n)


class Engine:
    """One GPU-resident graph shard (a bmx_ctx). Host-array methods are synchronous; *_dev methods take
    device tensors/pointers and only enqueue work on the engine's stream."""

    def __init__(self, capacity_rows, device=0, flags=0, load_pct=0):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.bmx_create_ex(int(device), int(capacity_rows), int(load_pct), int(flags) | DEFAULT_CTX_FLAGS, C.byref(h))
        if rc != OK:
            raise BmxError(rc, (self.L.bmx_last_error(None) or b"").decode())
        self.h = h
        self.device = device

    def _chk(self, rc):
        if rc < 0:
            raise BmxError(rc, (self.L.bmx_last_error(self.h) or b"").decode())
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.L.bmx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- host-array (synchronous) API ----
    def load_rows(self, id, field, ts, val):
        id, field, ts, val = _np(id, np.uint64), _np(field, np.uint32), _np(ts, np.int64), _np(val, np.int64)
        self._chk(self.L.bmx_load_rows(self.h, len(id), _ptr(id), _ptr(field), _ptr(ts), _ptr(val), MEM_HOST))

    def put_rows(self, id, field, ts, val):
        """rows decided elsewhere, stored as given (unique keys per call); val == VAL_DELETED leaves a tombstone"""
        id, field, ts, val = _np(id, np.uint64), _np(field, np.uint32), _np(ts, np.int64), _np(val, np.int64)
        self._chk(self.L.bmx_put_rows(self.h, len(id), _ptr(id), _ptr(field), _ptr(ts), _ptr(val), MEM_HOST))

    def merge_batch(self, id, field, ts, val, insert_mode=INSERT_REFERENCE, want_flags=True):
        """Returns (applied_idx u32[w] ascending, flags u8[n] or None, MergeStats)."""
        id, field, ts, val = _np(id, np.uint64), _np(field, np.uint32), _np(ts, np.int64), _np(val, np.int64)
        n = len(id)
        applied = np.zeros(max(n, 1), np.uint32)
        flags = np.zeros(max(n, 1), np.uint8) if want_flags else None
        na = C.c_uint64(0)
        st = MergeStats()
        self._chk(self.L.bmx_merge_batch(self.h, n, _ptr(id), _ptr(field), _ptr(ts), _ptr(val), int(insert_mode), MEM_HOST,
                                         _ptr(applied), C.cast(C.byref(na), C.c_void_p), _ptr(flags), C.cast(C.byref(st), C.c_void_p)))
        return applied[:na.value].copy(), (flags[:n] if want_flags else None), st

    def merge_submit(self, id, field, ts, val, insert_mode=INSERT_REFERENCE, want_flags=False):
        """Upload a host batch and enqueue its merge; -> ticket for merge_collect(). At most two batches in flight."""
        id, field, ts, val = _np(id, np.uint64), _np(field, np.uint32), _np(ts, np.int64), _np(val, np.int64)
        t = C.c_uint64()
        self._chk(self.L.bmx_merge_submit(self.h, len(id), _ptr(id), _ptr(field), _ptr(ts), _ptr(val), int(insert_mode), 1 if want_flags else 0, C.byref(t)))
        return (t.value, len(id), want_flags)

    def merge_collect(self, ticket):
        """-> (applied_idx u32[w], flags u8[n] or None, MergeStats) of a submitted batch (collect in submission order)."""
        t, n, want_flags = ticket
        applied = np.zeros(max(n, 1), np.uint32)
        flags = np.zeros(max(n, 1), np.uint8) if want_flags else None
        na = C.c_uint64(0)
        st = MergeStats()
        self._chk(self.L.bmx_merge_collect(self.h, int(t), _ptr(applied), C.cast(C.byref(na), C.c_void_p), _ptr(flags), C.cast(C.byref(st), C.c_void_p)))
        return applied[:na.value].copy(), (flags[:n] if want_flags else None), st

    def get_rows(self, id, field):
        id, field = _np(id, np.uint64), _np(field, np.uint32)
        n = len(id)
        ts = np.zeros(n, np.int64); val = np.zeros(n, np.int64); found = np.zeros(n, np.uint8)
        self._chk(self.L.bmx_get_rows(self.h, n, _ptr(id), _ptr(field), _ptr(ts), _ptr(val), _ptr(found), MEM_HOST))
        return ts, val, found.astype(bool)

    def get_row(self, id, field):
        ts, val = C.c_int64(), C.c_int64()
        rc = self._chk(self.L.bmx_get_row(self.h, int(id), int(field), C.byref(ts), C.byref(val)))
        return (ts.value, val.value) if rc == 1 else None

    def row_count(self):
        n = C.c_uint64()
        self._chk(self.L.bmx_row_count(self.h, C.byref(n)))
        return n.value

    def reserve(self, capacity_rows):
        self._chk(self.L.bmx_reserve(self.h, int(capacity_rows)))

    def dump_rows(self):
        n = self.row_count()
        id = np.zeros(n, np.uint64); field = np.zeros(n, np.uint32); ts = np.zeros(n, np.int64); val = np.zeros(n, np.int64)
        m = C.c_uint64()
        self._chk(self.L.bmx_dump_rows(self.h, n, _ptr(id), _ptr(field), _ptr(ts), _ptr(val), C.cast(C.byref(m), C.c_void_p), MEM_HOST))
        assert m.value <= n       # tombstones hold a slot (row_count) but are not dumped
        k = m.value
        return id[:k], field[:k], ts[:k], val[:k]

    def index_build(self, field):
        self._chk(self.L.bmx_index_build(self.h, int(field)))

    def index_drop(self, field):
        self._chk(self.L.bmx_index_drop(self.h, int(field)))

    def index_size(self, field):
        n = C.c_uint64()
        self._chk(self.L.bmx_index_size(self.h, int(field), C.byref(n)))
        return n.value

    def index_refresh_counts(self):
        """-> (full index builds, incremental updates from the change log) since the engine was created"""
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(self.L.bmx_index_refresh_counts(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def index_set_ordered(self, field, after_queries=1):
        """value-ordered view of the index (bmx_index_set_ordered): sorted again by the after_queries-th query since the field last changed; 0 = off"""
        self._chk(self.L.bmx_index_set_ordered(self.h, int(field), int(after_queries)))

    def index_ordered_info(self, field):
        """-> (after_queries, would the view answer the next query, sorts so far)"""
        a, v, n = C.c_uint32(), C.c_int32(), C.c_uint64()
        self._chk(self.L.bmx_index_ordered_info(self.h, int(field), C.byref(a), C.byref(v), C.byref(n)))
        return a.value, bool(v.value), n.value

    def index_ordered_stats(self, field):
        """{sorts, patches, keys_patched, last_sort_us, last_patch_us, rewrites (streaming merges into the view's main run), pending_keys} of the value-ordered view"""
        a, b, c, f, g = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        d, e = C.c_double(), C.c_double()
        self._chk(self.L.bmx_index_ordered_stats(self.h, int(field), C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e), C.byref(f), C.byref(g)))
        return {"sorts": a.value, "patches": b.value, "keys_patched": c.value, "last_sort_us": d.value, "last_patch_us": e.value, "rewrites": f.value, "pending_keys": g.value}

    def scan_range(self, field, lo, hi, cap=None, out=None):
        """ids of the rows whose value is in [lo, hi]. `out`: a caller's uint64 array the ids are written into (a view of it is returned, no copy) —
        one taken from a HostBuffer comes back at the link's rate."""
        if out is not None:
            cap = len(out) if cap is None else min(cap, len(out))
            m = C.c_uint64()
            self._chk(self.L.bmx_scan_range(self.h, int(field), int(lo), int(hi), _ptr(out), cap, C.cast(C.byref(m), C.c_void_p), MEM_HOST))
            return out[:min(m.value, cap)]
        cap = self.index_size(field) if cap is None else cap
        out = np.zeros(max(cap, 1), np.uint64)
        m = C.c_uint64()
        self._chk(self.L.bmx_scan_range(self.h, int(field), int(lo), int(hi), _ptr(out), cap, C.cast(C.byref(m), C.c_void_p), MEM_HOST))
        return out[:min(m.value, cap)].copy()

    def scan_range_pos(self, field, lo, hi, cap=None):
        """positions (u32, ascending) of the matches in the index columns instead of their node ids"""
        cap = self.index_size(field) if cap is None else cap
        out = np.zeros(max(cap, 1), np.uint32)
        m = C.c_uint64()
        self._chk(self.L.bmx_scan_range_pos(self.h, int(field), int(lo), int(hi), _ptr(out), cap, C.cast(C.byref(m), C.c_void_p), MEM_HOST))
        return out[:min(m.value, cap)].copy()

    def index_ids(self, field, first=0, count=None):
        count = self.index_size(field) - first if count is None else count
        out = np.zeros(max(count, 1), np.uint64)
        self._chk(self.L.bmx_index_ids(self.h, int(field), int(first), int(count), _ptr(out), MEM_HOST))
        return out[:count].copy()

    def scan_equals(self, field, value):
        return self.scan_range(field, value, value)

    def scan_count(self, field, lo, hi):
        m = C.c_uint64()
        self._chk(self.L.bmx_scan_count(self.h, int(field), int(lo), int(hi), C.cast(C.byref(m), C.c_void_p), MEM_HOST))
        return m.value

    def scan_filter(self, terms, cap=None):
        arr = (Term * len(terms))(*[Term(int(f), 0, int(lo), int(hi)) for f, lo, hi in terms])
        cap = self.index_size(terms[0][0]) if cap is None else cap
        out = np.zeros(max(cap, 1), np.uint64)
        m = C.c_uint64()
        self._chk(self.L.bmx_scan_filter(self.h, len(terms), arr, _ptr(out), cap, C.cast(C.byref(m), C.c_void_p), MEM_HOST))
        return out[:min(m.value, cap)].copy()

    def info(self):
        i = Info()
        self._chk(self.L.bmx_get_info(self.h, C.byref(i)))
        return i

    # ---- device-pointer (asynchronous) API: arguments are torch CUDA tensors or raw device addresses ----
    def sync(self):
        self._chk(self.L.bmx_sync(self.h))

    def set_stream(self, stream_ptr):
        self._chk(self.L.bmx_set_stream(self.h, C.c_void_p(stream_ptr) if stream_ptr else None))

    def set_deferred(self, on):
        """deferred compaction (bmx.h): the winner compaction of a device batch runs under the NEXT batch's probe kernel. On by default."""
        self._chk(self.L.bmx_set_deferred_compaction(self.h, 1 if on else 0))

    def set_side_stream(self, stream_ptr):
        """the stream the deferred compactions run on (0 / None: the context's own high-priority stream)"""
        self._chk(self.L.bmx_set_side_stream(self.h, C.c_void_p(stream_ptr) if stream_ptr else None))

    def set_wait_limit(self, seconds):
        """device-side waits on this context's GPU give up after this long (default ~60 s): bmx_set_wait_limit"""
        self._chk(self.L.bmx_set_wait_limit(self.h, float(seconds)))

    def merge_fence(self):
        """enqueue-only: the engine's stream is ordered behind every compaction (for work the caller enqueues on that stream itself)"""
        self._chk(self.L.bmx_merge_fence(self.h))

    def set_probe_waves(self, waves_per_simd):
        """resident waves per SIMD the probe kernel may take (8, 6, 5, 4, 3): fewer leave room for kernels that run beside it (bmx_set_probe_waves)"""
        self._chk(self.L.bmx_set_probe_waves(self.h, int(waves_per_simd)))

    def placement(self):
        """table placement tuning at create / growth: {candidates, probe_us_chosen, probe_us_slowest} (candidates 0: not tuned)"""
        n, a, b = C.c_uint32(), C.c_float(), C.c_float()
        self._chk(self.L.bmx_get_placement(self.h, C.byref(n), C.byref(a), C.byref(b)))
        return {"candidates": n.value, "probe_us_chosen": round(a.value, 2), "probe_us_slowest": round(b.value, 2)}

    def deferred_counts(self):
        """(merges whose compaction was deferred, of those: run on the side stream under the next probe kernel)"""
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(self.L.bmx_get_deferred_counts(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def seq_signal(self, stream_ptr, seq_dev, value):
        """enqueue on the stream (0 = the engine's): *seq_dev = value once everything before it on that stream is done."""
        self._chk(self.L.bmx_seq_signal(self.h, C.c_void_p(stream_ptr) if stream_ptr else None, _ptr(seq_dev), int(value)))

    def seq_wait(self, stream_ptr, seq_dev, at_least):
        """enqueue on the stream a device-side wait until *seq_dev >= at_least (the matching signal must already be enqueued)."""
        self._chk(self.L.bmx_seq_wait(self.h, C.c_void_p(stream_ptr) if stream_ptr else None, _ptr(seq_dev), int(at_least)))

    def load_rows_dev(self, n, id, field, ts, val):
        self._chk(self.L.bmx_load_rows(self.h, int(n), _ptr(id), _ptr(field), _ptr(ts), _ptr(val), MEM_DEVICE))

    def merge_batch_dev(self, n, id, field, ts, val, insert_mode=INSERT_REFERENCE, applied=None, n_applied=None, flags=None, stats=None):
        self._chk(self.L.bmx_merge_batch(self.h, int(n), _ptr(id), _ptr(field), _ptr(ts), _ptr(val), int(insert_mode), MEM_DEVICE,
                                         _ptr(applied), _ptr(n_applied), _ptr(flags), _ptr(stats)))

    def merge_records_dev(self, n, recs, insert_mode=INSERT_REFERENCE, applied=None, n_applied=None, flags=None, stats=None):
        self._chk(self.L.bmx_merge_records(self.h, int(n), _ptr(recs), int(insert_mode), _ptr(applied), _ptr(n_applied), _ptr(flags), _ptr(stats)))

    def partition_by_owner_dev(self, n, id, field, ts, val, nshards, recs_out, counts_out):
        self._chk(self.L.bmx_partition_by_owner(self.h, int(n), _ptr(id), _ptr(field), _ptr(ts), _ptr(val), int(nshards), _ptr(recs_out), _ptr(counts_out)))

    # ---- direct exchange between processes (bmx_ipc_* / bmx_partition_scatter): pointers are plain ints here ----
    def ipc_alloc(self, nbytes, uncached=True):
        """-> (device pointer, 64-byte handle another process opens with ipc_open). uncached: never served from this GPU's L2 (peers store into it)"""
        p = C.c_void_p()
        h = C.create_string_buffer(64)
        self._chk(self.L.bmx_ipc_alloc(self.h, int(nbytes), 1 if uncached else 0, C.byref(p), h))
        return int(p.value), h.raw

    def ipc_open(self, handle, peer_device=-1):
        p = C.c_void_p()
        self._chk(self.L.bmx_ipc_open(self.h, C.create_string_buffer(bytes(handle), 64), int(peer_device), C.byref(p)))
        return int(p.value)

    def ipc_close(self, ptr):
        self._chk(self.L.bmx_ipc_close(self.h, C.c_void_p(int(ptr))))

    def ipc_free(self, ptr):
        self._chk(self.L.bmx_ipc_free(self.h, C.c_void_p(int(ptr))))

    def partition_scatter_dev(self, n, id, field, ts, val, nshards, slab_records, dst_ptrs, counts_out, arrive_ptrs=None, arrive_value=0):
        """owner partition writing slab g straight to dst_ptrs[g] (this GPU's or a peer's memory); arrive_ptrs[g] := arrive_value when all is stored"""
        dst = (C.c_void_p * int(nshards))(*[C.c_void_p(int(x)) for x in dst_ptrs])
        arr = (C.c_void_p * int(nshards))(*[C.c_void_p(int(x)) if x else None for x in arrive_ptrs]) if arrive_ptrs is not None else None
        self._chk(self.L.bmx_partition_scatter(self.h, int(n), _ptr(id), _ptr(field), _ptr(ts), _ptr(val), int(nshards), int(slab_records), dst, _ptr(counts_out),
                                               arr, int(arrive_value), None, 0, 0))

    @staticmethod
    def ptr_array(ptrs):
        """a C array of device pointers, built once and reused (partition_scatter_raw): keeps the per-step host cost down"""
        return (C.c_void_p * max(len(ptrs), 1))(*[C.c_void_p(int(x)) if x else None for x in ptrs])

    def partition_scatter_raw(self, n, id, field, ts, val, nshards, slab_records, dst_arr, counts_out, arrive_arr, arrive_value, wait_ptr=0, n_wait=0, wait_at_least=0):
        """one host call per route: (optional) wait until n_wait words at wait_ptr are >= wait_at_least, partition + scatter, arrival words"""
        self._chk(self.L.bmx_partition_scatter(self.h, n, _ptr(id), _ptr(field), _ptr(ts), _ptr(val), nshards, slab_records, dst_arr, _ptr(counts_out),
                                               arrive_arr, arrive_value, wait_ptr if wait_ptr else None, n_wait, wait_at_least))

    def merge_records_after(self, wait_ptr, n_wait, wait_at_least, n, recs_ptr, insert_mode, applied, n_applied):
        """one host call per merge: wait for the arrival words, then merge the records"""
        self._chk(self.L.bmx_merge_records_after(self.h, wait_ptr, n_wait, wait_at_least, n, recs_ptr, insert_mode, _ptr(applied), _ptr(n_applied), None, None))

    def seq_wait_all(self, stream_ptr, words_ptr, nwords, at_least):
        self._chk(self.L.bmx_seq_wait_all(self.h, C.c_void_p(stream_ptr) if stream_ptr else None, _ptr(words_ptr), int(nwords), int(at_least)))

    def merge_tail_wait(self, words_ptr, nwords, at_least):
        """the NEXT merge's resolve kernel ends only once the words are >= at_least (bmx_merge_tail_wait); 0 words disarms"""
        self._chk(self.L.bmx_merge_tail_wait(self.h, C.c_void_p(int(words_ptr)) if words_ptr else None, int(nwords), int(at_least)))

    def merge_notify(self, word_ptrs):
        """every later merge stores the count of merges finished since into these words (peers' memory); [] switches it off"""
        n = len(word_ptrs)
        arr = (C.c_void_p * max(n, 1))(*[C.c_void_p(int(x)) if x else None for x in word_ptrs]) if n else None
        self._chk(self.L.bmx_merge_notify(self.h, arr, n))

    def partition_by_owner_slabs_dev(self, n, id, field, ts, val, nshards, slab_records, recs_out, counts_out):
        self._chk(self.L.bmx_partition_by_owner_slabs(self.h, int(n), _ptr(id), _ptr(field), _ptr(ts), _ptr(val), int(nshards), int(slab_records),
                                                      _ptr(recs_out), _ptr(counts_out)))

    def scan_range_dev(self, field, lo, hi, out_ids, cap, n_out):
        self._chk(self.L.bmx_scan_range(self.h, int(field), int(lo), int(hi), _ptr(out_ids), int(cap), _ptr(n_out), MEM_DEVICE))

    def scan_range_pos_dev(self, field, lo, hi, out_pos, cap, n_out):
        self._chk(self.L.bmx_scan_range_pos(self.h, int(field), int(lo), int(hi), _ptr(out_pos), int(cap), _ptr(n_out), MEM_DEVICE))

    def index_ids_dev(self, field, first, count, out_ids):
        self._chk(self.L.bmx_index_ids(self.h, int(field), int(first), int(count), _ptr(out_ids), MEM_DEVICE))

    def profile_enable(self, on=True):
        self._chk(self.L.bmx_profile_enable(self.h, 1 if on else 0))

    def profile_read(self):
        """-> (dict stage -> average ms per merge call, number of calls)"""
        ms = (C.c_float * 3)()
        n = C.c_uint32()
        self._chk(self.L.bmx_profile_read(self.h, ms, C.byref(n)))
        return {"probe_apply": ms[0], "resolve_lists": ms[1], "compact": ms[2]}, n.value

    def profile_read_scan(self):
        """-> (dict stage -> average ms per scan call, number of calls)"""
        ms = (C.c_float * 2)()
        n = C.c_uint32()
        self._chk(self.L.bmx_profile_read_scan(self.h, ms, C.byref(n)))
        return {"scan_mask": ms[0], "emit": ms[1]}, n.value

    def timer_start(self):
        self._chk(self.L.bmx_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self._chk(self.L.bmx_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def timer_mark(self):
        """record the stop event now (enqueue only; the last batch's deferred compaction is launched in front of it)"""
        self._chk(self.L.bmx_timer_mark(self.h))

    def timer_elapsed(self):
        ms = C.c_float()
        self._chk(self.L.bmx_timer_elapsed(self.h, C.byref(ms)))
        return ms.value


class Comm:
    """N shards in one process (a bmx_comm): the graph split by node-id hash over N contexts. devices may repeat a GPU (logical shards)."""

    def __init__(self, devices, capacity_rows_per_shard, flags=0):
        self.L = load_library()
        self.N = len(devices)
        arr = (C.c_int * self.N)(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = self.L.bmx_comm_create(self.N, arr, int(capacity_rows_per_shard), int(flags) | DEFAULT_CTX_FLAGS, C.byref(h))
        if rc != OK:
            raise BmxError(rc, (self.L.bmx_comm_last_error(None) or b"").decode())
        self.h = h

    def _chk(self, rc):
        if rc < 0:
            raise BmxError(rc, (self.L.bmx_comm_last_error(self.h) or b"").decode())
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.L.bmx_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        self._chk(self.L.bmx_comm_sync(self.h))

    def load_rows(self, id, field, ts, val):
        id, field, ts, val = _np(id, np.uint64), _np(field, np.uint32), _np(ts, np.int64), _np(val, np.int64)
        self._chk(self.L.bmx_comm_load_rows(self.h, len(id), _ptr(id), _ptr(field), _ptr(ts), _ptr(val)))

    def put_rows(self, id, field, ts, val):
        id, field, ts, val = _np(id, np.uint64), _np(field, np.uint32), _np(ts, np.int64), _np(val, np.int64)
        self._chk(self.L.bmx_comm_put_rows(self.h, len(id), _ptr(id), _ptr(field), _ptr(ts), _ptr(val)))

    def merge(self, id, field, ts, val, insert_mode=INSERT_REFERENCE, applied_out=None):
        """host batch -> (applied_idx u32[w] ascending indices into the batch, MergeStats summed over the shards).
        applied_out: a uint32 array of at least n entries to receive the winners (e.g. page-locked: HostBuffer); the result is then a view of it."""
        id, field, ts, val = _np(id, np.uint64), _np(field, np.uint32), _np(ts, np.int64), _np(val, np.int64)
        n = len(id)
        if applied_out is not None and (applied_out.dtype != np.uint32 or len(applied_out) < n or not applied_out.flags["C_CONTIGUOUS"]):
            raise ValueError("applied_out must be a contiguous uint32 array of at least n entries")
        applied = applied_out if applied_out is not None else np.zeros(max(n, 1), np.uint32)
        na = C.c_uint64(0)
        st = MergeStats()
        self._chk(self.L.bmx_comm_merge(self.h, n, _ptr(id), _ptr(field), _ptr(ts), _ptr(val), int(insert_mode), _ptr(applied),
                                        C.cast(C.byref(na), C.c_void_p), C.cast(C.byref(st), C.c_void_p)))
        return (applied[:na.value] if applied_out is not None else applied[:na.value].copy()), st

    def merge_dev(self, batches, insert_mode=INSERT_REFERENCE, slab_records=0):
        """batches[i] = (n, id, field, ts, val): device tensors on shard i's GPU, the deltas shard i originates. Enqueue-only."""
        N = self.N
        assert len(batches) == N
        ns = (C.c_uint64 * N)(*[int(b[0]) for b in batches])
        cols = [(C.c_void_p * N)(*[(b[1 + k].data_ptr() if b[0] else 0) for b in batches]) for k in range(4)]
        self._chk(self.L.bmx_comm_merge_dev(self.h, ns, cols[0], cols[1], cols[2], cols[3], int(insert_mode), int(slab_records)))

    def shard_result(self, g):
        """-> (recs_ptr, applied_ptr, n_applied_ptr, stats_ptr, n_records): raw device addresses of shard g's last device step"""
        r, a, n, s = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        m = C.c_uint64()
        self._chk(self.L.bmx_comm_shard_result(self.h, int(g), C.byref(r), C.byref(a), C.byref(n), C.byref(s), C.byref(m)))
        return r.value, a.value, n.value, s.value, m.value

    def row_count(self):
        n = C.c_uint64()
        self._chk(self.L.bmx_comm_row_count(self.h, C.byref(n)))
        return n.value

    def get_rows(self, id, field):
        id, field = _np(id, np.uint64), _np(field, np.uint32)
        n = len(id)
        ts = np.zeros(n, np.int64); val = np.zeros(n, np.int64); found = np.zeros(n, np.uint8)
        self._chk(self.L.bmx_comm_get_rows(self.h, n, _ptr(id), _ptr(field), _ptr(ts), _ptr(val), _ptr(found)))
        return ts, val, found.astype(bool)

    def dump_rows(self):
        n = self.row_count()
        id = np.zeros(n, np.uint64); field = np.zeros(n, np.uint32); ts = np.zeros(n, np.int64); val = np.zeros(n, np.int64)
        m = C.c_uint64()
        self._chk(self.L.bmx_comm_dump_rows(self.h, n, _ptr(id), _ptr(field), _ptr(ts), _ptr(val), C.byref(m)))
        assert m.value <= n
        k = m.value
        return id[:k], field[:k], ts[:k], val[:k]

    def index_build(self, field):
        self._chk(self.L.bmx_comm_index_build(self.h, int(field)))

    def index_set_ordered(self, field, after_queries=1):
        self._chk(self.L.bmx_comm_index_set_ordered(self.h, int(field), int(after_queries)))

    def scan_count(self, field, lo, hi):
        m = C.c_uint64()
        self._chk(self.L.bmx_comm_scan_count(self.h, int(field), int(lo), int(hi), C.byref(m)))
        return m.value

    def scan_range(self, field, lo, hi):
        cap = self.scan_count(field, lo, hi)
        out = np.zeros(max(cap, 1), np.uint64)
        m = C.c_uint64()
        self._chk(self.L.bmx_comm_scan_range(self.h, int(field), int(lo), int(hi), _ptr(out), cap, C.byref(m)))
        return out[:min(m.value, cap)].copy()

    def scan_equals(self, field, value):
        return self.scan_range(field, value, value)

    def scan_filter(self, terms):
        arr = (Term * len(terms))(*[Term(int(f), 0, int(lo), int(hi)) for f, lo, hi in terms])
        m = C.c_uint64()
        self._chk(self.L.bmx_comm_scan_filter(self.h, len(terms), arr, None, 0, C.byref(m)))
        cap = m.value
        out = np.zeros(max(cap, 1), np.uint64)
        self._chk(self.L.bmx_comm_scan_filter(self.h, len(terms), arr, _ptr(out), cap, C.byref(m)))
        return out[:min(m.value, cap)].copy()


FLAG_CONCURRENT = 8
VC_ABSENT, VC_DENSE, VC_SPARSE = 0, 1, 2


class EngineVC:
    """N4 table (a bmx_vc): rows carry a K-writer vector clock instead of one timestamp. Host arrays only, synchronous.
    clocks are (n, K) uint32, component k = writer k's counter (0 = absent from the reference's clock object)."""

    def __init__(self, capacity_rows, k_writers, local_writer, device=0):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.bmx_vc_create(int(device), int(capacity_rows), int(k_writers), int(local_writer), C.byref(h))
        if rc != OK:
            raise BmxError(rc, (self.L.bmx_vc_last_error(None) or b"").decode())
        self.h = h
        self.K = int(k_writers)

    def _chk(self, rc):
        if rc < 0:
            raise BmxError(rc, (self.L.bmx_vc_last_error(self.h) or b"").decode())
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.L.bmx_vc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _cols(self, id, field, clocks, val):
        id = _np(id, np.uint64); field = _np(field, np.uint32); val = _np(val, np.int64)
        clocks = _np(clocks, np.uint32).reshape(len(id), self.K) if len(id) else np.zeros((0, self.K), np.uint32)
        if not (len(id) == len(field) == len(val)):
            raise ValueError("column lengths differ")
        return id, field, clocks, val

    def load_rows(self, id, field, clocks, val, keysets=None):
        """keysets (uint32[n], see keyset()): which writers each clock names and in which order; None = all K, in order."""
        id, field, clocks, val = self._cols(id, field, clocks, val)
        ks = None if keysets is None else _np(keysets, np.uint32)
        self._chk(self.L.bmx_vc_load_rows_ks(self.h, len(id), _ptr(id), _ptr(field), _ptr(clocks), _ptr(ks), _ptr(val)))

    def merge_batch(self, id, field, clocks, val, keysets=None):
        """-> (flags uint8[n], updated uint32[]): per-delta resolve() flags; for each touched row that changed, the index of the
        last delta that updated it, ascending."""
        id, field, clocks, val = self._cols(id, field, clocks, val)
        n = len(id)
        ks = None if keysets is None else _np(keysets, np.uint32)
        if ks is not None and len(ks) != n:
            raise ValueError("column lengths differ")
        flags = np.zeros(n, np.uint8); upd = np.zeros(max(n, 1), np.uint32); nu = C.c_uint64()
        self._chk(self.L.bmx_vc_merge_batch_ks(self.h, n, _ptr(id), _ptr(field), _ptr(clocks), _ptr(ks), _ptr(val), _ptr(upd), C.byref(nu), _ptr(flags)))
        return flags, upd[: nu.value].copy()

    def get_rows(self, id, field, with_keysets=False):
        """-> (clocks (n,K) uint32, val int64[n], state uint8[n]) with state VC_ABSENT / VC_DENSE / VC_SPARSE; with_keysets: + keysets uint32[n]."""
        id = _np(id, np.uint64); field = _np(field, np.uint32)
        n = len(id)
        clocks = np.zeros((n, self.K), np.uint32); val = np.zeros(n, np.int64); st = np.zeros(n, np.uint8)
        if with_keysets:
            ks = np.zeros(n, np.uint32)
            self._chk(self.L.bmx_vc_get_rows_ks(self.h, n, _ptr(id), _ptr(field), _ptr(clocks), _ptr(ks), _ptr(val), _ptr(st)))
            return clocks, val, st, ks
        self._chk(self.L.bmx_vc_get_rows(self.h, n, _ptr(id), _ptr(field), _ptr(clocks), _ptr(val), _ptr(st)))
        return clocks, val, st

    def row_count(self):
        n = C.c_uint64()
        self._chk(self.L.bmx_vc_row_count(self.h, C.byref(n)))
        return n.value

    def scan_range(self, field, lo, hi, count_only=False):
        """node ids of the rows of `field` with lo <= val <= hi (table order), or their number"""
        m = C.c_uint64()
        if count_only:
            self._chk(self.L.bmx_vc_scan_range(self.h, int(field), int(lo), int(hi), None, 0, C.byref(m)))
            return m.value
        cap = max(self.row_count(), 1)
        out = np.zeros(cap, np.uint64)
        self._chk(self.L.bmx_vc_scan_range(self.h, int(field), int(lo), int(hi), _ptr(out), cap, C.byref(m)))
        return out[:min(m.value, cap)].copy()

    # device-pointer form: torch tensors / raw pointers, enqueue-only
    def merge_batch_dev(self, n, id, field, clocks, val, updated=None, n_updated=None, flags=None, keysets=None):
        self._chk(self.L.bmx_vc_merge_batch_ks_dev(self.h, int(n), _ptr(id), _ptr(field), _ptr(clocks), _ptr(keysets), _ptr(val), _ptr(updated), _ptr(n_updated), _ptr(flags)))

    def set_stream(self, stream_ptr):
        self._chk(self.L.bmx_vc_set_stream(self.h, C.c_void_p(stream_ptr) if stream_ptr else None))

    def sync(self):
        self._chk(self.L.bmx_vc_sync(self.h))


KEYSET_NONE = 0xFFFFFFFF


def keyset(writers):
    """Key-set word of a clock whose keys are the writers with these indices, in this order (include/bmx.h bmx_vc_keyset)."""
    ks = KEYSET_NONE
    for i, w in enumerate(writers):
        ks = (ks & ~(0xF << (4 * i))) | ((int(w) & 0xF) << (4 * i))
    return ks & 0xFFFFFFFF


def keyset_writers(ks):
    """inverse of keyset(): the writer indices a key-set word names, in order"""
    out = []
    for i in range(8):
        w = (int(ks) >> (4 * i)) & 0xF
        if w == 0xF:
            break
        out.append(w)
    return out


def owner_of(ids, nshards):
    L = load_library()
    return np.array([L.bmx_owner_of(int(i), int(nshards)) for i in np.asarray(ids, dtype=np.uint64)], dtype=np.uint32)
