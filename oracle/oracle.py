"""ctypes/numpy wrapper over oracle/libbmx_oracle.so — TEST INFRASTRUCTURE ONLY.

The C file restates the reference's scalar-clock merge (src/bullet-crt.js:164-279) and fresh-index
scans (src/bullet-query.js:186-313); this wrapper only marshals numpy arrays.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libbmx_oracle.so")

FLAG_INCOMING, FLAG_CURRENT, FLAG_HISTORICAL = 1, 2, 4
INSERT_REFERENCE, INSERT_DELTA = 0, 1
VAL_DELETED = -(1 << 63)   # ORC_VAL_DELETED: tombstone value


def build(force=False):
    src = os.path.join(_HERE, "bmx_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libbmx_oracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        u64p, u32p, i64p, u8p = (C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8))
        L.orc_create.restype = C.c_void_p
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_size.argtypes = [C.c_void_p]; L.orc_size.restype = C.c_uint64
        L.orc_load_rows.argtypes = [C.c_void_p, C.c_uint64, u64p, u32p, i64p, i64p]
        L.orc_merge_batch.argtypes = [C.c_void_p, C.c_uint64, u64p, u32p, i64p, i64p, C.c_int, u8p, u32p]
        L.orc_merge_batch_marked.argtypes = [C.c_void_p, C.c_uint64, u64p, u32p, i64p, i64p, C.c_int, u8p, u32p, u8p]; L.orc_merge_batch_marked.restype = C.c_uint64
        L.orc_merge_batch.restype = C.c_uint64
        L.orc_get_row.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, i64p, i64p]; L.orc_get_row.restype = C.c_int
        L.orc_dump_rows.argtypes = [C.c_void_p, C.c_uint64, u64p, u32p, i64p, i64p]; L.orc_dump_rows.restype = C.c_uint64
        L.orc_digest.argtypes = [C.c_void_p]; L.orc_digest.restype = C.c_uint64
        L.orc_row_digest.argtypes = [C.c_uint64, C.c_uint32, C.c_int64, C.c_int64]; L.orc_row_digest.restype = C.c_uint64
        L.orc_scan_range.argtypes = [C.c_void_p, C.c_uint32, C.c_int64, C.c_int64, u64p, C.c_uint64]; L.orc_scan_range.restype = C.c_uint64
        L.orc_scan_filter_and.argtypes = [C.c_void_p, C.c_uint32, u32p, i64p, i64p, u64p, C.c_uint64]
        L.orc_scan_filter_and.restype = C.c_uint64
        L.orc_owner_of.argtypes = [C.c_uint64, C.c_uint32]; L.orc_owner_of.restype = C.c_uint32
        L.orc_mt_batch.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint64, u64p, u32p, i64p, i64p, C.c_int, C.c_int]; L.orc_mt_batch.restype = C.c_uint64
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _cols(id, field, ts, val):
    return (np.ascontiguousarray(id, dtype=np.uint64), np.ascontiguousarray(field, dtype=np.uint32),
            np.ascontiguousarray(ts, dtype=np.int64), np.ascontiguousarray(val, dtype=np.int64))


class Oracle:
    """Sequential CPU table with the reference's scalar-clock merge semantics."""

    def __init__(self):
        self._L = lib()
        self._h = C.c_void_p(self._L.orc_create())

    def close(self):
        if self._h:
            self._L.orc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self._L.orc_size(self._h))

    def load_rows(self, id, field, ts, val):
        id, field, ts, val = _cols(id, field, ts, val)
        self._L.orc_load_rows(self._h, len(id), _p(id, C.c_uint64), _p(field, C.c_uint32), _p(ts, C.c_int64), _p(val, C.c_int64))

    def merge_batch(self, id, field, ts, val, insert_mode=INSERT_REFERENCE):
        """Returns (flags u8[n], winners u32[w] ascending)."""
        id, field, ts, val = _cols(id, field, ts, val)
        n = len(id)
        flags = np.zeros(n, dtype=np.uint8)
        winners = np.zeros(max(n, 1), dtype=np.uint32)
        w = self._L.orc_merge_batch(self._h, n, _p(id, C.c_uint64), _p(field, C.c_uint32), _p(ts, C.c_int64), _p(val, C.c_int64),
                                    int(insert_mode), _p(flags, C.c_uint8), _p(winners, C.c_uint32))
        return flags, winners[:w].copy()

    def merge_batch_marked(self, id, field, ts, val, insert_mode=INSERT_REFERENCE):
        """winners u32[w] ascending with bit 31 set on the winners that created their row (what BMX_MERGE_MARK_CREATED reports)"""
        id, field, ts, val = _cols(id, field, ts, val)
        n = len(id)
        winners = np.zeros(max(n, 1), dtype=np.uint32); created = np.zeros(max(n, 1), dtype=np.uint8)
        w = self._L.orc_merge_batch_marked(self._h, n, _p(id, C.c_uint64), _p(field, C.c_uint32), _p(ts, C.c_int64), _p(val, C.c_int64),
                                           int(insert_mode), None, _p(winners, C.c_uint32), _p(created, C.c_uint8))
        win = winners[:w].copy()
        return win | (created[win].astype(np.uint32) << np.uint32(31))

    def get_row(self, id, field):
        ts, val = C.c_int64(), C.c_int64()
        ok = self._L.orc_get_row(self._h, int(id), int(field), C.byref(ts), C.byref(val))
        return (ts.value, val.value) if ok else None

    put_rows = load_rows     # rows decided elsewhere, stored as given (val == VAL_DELETED: tombstone): the restatement of bmx_put_rows

    def dump_rows(self):
        n = len(self)
        id = np.zeros(n, np.uint64); field = np.zeros(n, np.uint32); ts = np.zeros(n, np.int64); val = np.zeros(n, np.int64)
        m = int(self._L.orc_dump_rows(self._h, n, _p(id, C.c_uint64), _p(field, C.c_uint32), _p(ts, C.c_int64), _p(val, C.c_int64)))   # tombstones are not dumped
        return id[:m], field[:m], ts[:m], val[:m]

    def digest(self):
        return int(self._L.orc_digest(self._h))

    def scan_range(self, field, lo, hi):
        cap = len(self)
        out = np.zeros(max(cap, 1), np.uint64)
        m = self._L.orc_scan_range(self._h, int(field), int(lo), int(hi), _p(out, C.c_uint64), cap)
        return out[:m].copy()

    def scan_equals(self, field, c):
        return self.scan_range(field, c, c)

    def scan_count(self, field, lo, hi):
        return int(self._L.orc_scan_range(self._h, int(field), int(lo), int(hi), None, 0))

    def scan_filter_and(self, terms):
        """terms: [(field, lo, hi), ...] all on the same node id."""
        k = len(terms)
        f = np.array([t[0] for t in terms], np.uint32); lo = np.array([t[1] for t in terms], np.int64); hi = np.array([t[2] for t in terms], np.int64)
        cap = len(self)
        out = np.zeros(max(cap, 1), np.uint64)
        m = self._L.orc_scan_filter_and(self._h, k, _p(f, C.c_uint32), _p(lo, C.c_int64), _p(hi, C.c_int64), _p(out, C.c_uint64), cap)
        return out[:m].copy()


def owner_of(ids, nshards):
    L = lib()
    return np.array([L.orc_owner_of(int(i), int(nshards)) for i in np.asarray(ids, dtype=np.uint64)], dtype=np.uint32)


class OracleMT:
    """All-cores variant (bench.py's extra CPU baseline line): T tables, thread k owns the keys with owner_of(id, T) == k."""

    def __init__(self, threads):
        self._L = lib()
        self.T = int(threads)
        self.tabs = [Oracle() for _ in range(self.T)]
        self._arr = (C.c_void_p * self.T)(*[t._h for t in self.tabs])

    def _run(self, cols, mode, load):
        id, field, ts, val = _cols(*cols)
        return int(self._L.orc_mt_batch(self._arr, self.T, len(id), _p(id, C.c_uint64), _p(field, C.c_uint32), _p(ts, C.c_int64), _p(val, C.c_int64),
                                        int(mode), int(load)))

    def load_rows(self, id, field, ts, val):
        self._run((id, field, ts, val), INSERT_REFERENCE, 1)

    def merge_batch(self, id, field, ts, val, insert_mode=INSERT_REFERENCE):
        """-> number of applied deltas"""
        return self._run((id, field, ts, val), insert_mode, 0)

    def __len__(self):
        return sum(len(t) for t in self.tabs)

    def digest(self):
        return sum(t.digest() for t in self.tabs) & ((1 << 64) - 1)

    def close(self):
        for t in self.tabs:
            t.close()


def rows_digest(id, field, ts, val):
    """Order-independent digest of a row set (numpy, vectorised); equals Oracle.digest() on the same rows."""
    with np.errstate(over="ignore"):
        def sm(x):
            z = (x + np.uint64(0x9e3779b97f4a7c15))
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
            return z ^ (z >> np.uint64(31))
        h = sm(np.asarray(val, np.int64).astype(np.uint64))
        h = sm(h ^ np.asarray(ts, np.int64).astype(np.uint64))
        h = sm(h ^ np.asarray(field, np.uint32).astype(np.uint64))
        h = sm(h ^ np.asarray(id, np.uint64))
        return int(h.sum(dtype=np.uint64))


FLAG_CONCURRENT = 8


class OracleVC:
    """N4: sequential CPU table with the reference's general (K-writer, dense) vector-clock merge semantics."""

    def __init__(self, K, local):
        L = lib()
        u64p, u32p, i64p, u8p = (C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8))
        L.orc_vc_create.restype = C.c_void_p; L.orc_vc_create.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_vc_destroy.argtypes = [C.c_void_p]
        L.orc_vc_size.argtypes = [C.c_void_p]; L.orc_vc_size.restype = C.c_uint64
        L.orc_vc_load_rows.argtypes = [C.c_void_p, C.c_uint64, u64p, u32p, u32p, i64p]
        L.orc_vc_merge_batch.argtypes = [C.c_void_p, C.c_uint64, u64p, u32p, u32p, i64p, u8p, u32p]; L.orc_vc_merge_batch.restype = C.c_uint64
        L.orc_vc_get_row.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, u32p, C.POINTER(C.c_int64), C.POINTER(C.c_int)]; L.orc_vc_get_row.restype = C.c_int
        L.orc_vc_dump_rows.argtypes = [C.c_void_p, C.c_uint64, u64p, u32p, u32p, i64p]; L.orc_vc_dump_rows.restype = C.c_uint64
        L.orc_vc_load_rows_ks.argtypes = [C.c_void_p, C.c_uint64, u64p, u32p, u32p, u32p, i64p]
        L.orc_vc_merge_batch_ks.argtypes = [C.c_void_p, C.c_uint64, u64p, u32p, u32p, u32p, i64p, u8p, u32p]; L.orc_vc_merge_batch_ks.restype = C.c_uint64
        L.orc_vc_get_row_ks.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, u32p, u32p, C.POINTER(C.c_int64), C.POINTER(C.c_int)]; L.orc_vc_get_row_ks.restype = C.c_int
        self._L, self.K = L, K
        self._h = C.c_void_p(L.orc_vc_create(K, local))
        assert self._h

    def __del__(self):
        try:
            if self._h:
                self._L.orc_vc_destroy(self._h); self._h = None
        except Exception:
            pass

    def __len__(self):
        return int(self._L.orc_vc_size(self._h))

    def _args(self, id, field, clocks, val):
        id = np.ascontiguousarray(id, np.uint64); field = np.ascontiguousarray(field, np.uint32)
        clocks = np.ascontiguousarray(clocks, np.uint32).reshape(len(id), self.K); val = np.ascontiguousarray(val, np.int64)
        return id, field, clocks, val

    def load_rows(self, id, field, clocks, val, keysets=None):
        id, field, clocks, val = self._args(id, field, clocks, val)
        ks = None if keysets is None else np.ascontiguousarray(keysets, np.uint32)
        self._L.orc_vc_load_rows_ks(self._h, len(id), _p(id, C.c_uint64), _p(field, C.c_uint32), _p(clocks, C.c_uint32),
                                    None if ks is None else _p(ks, C.c_uint32), _p(val, C.c_int64))

    def merge_batch(self, id, field, clocks, val, keysets=None):
        """-> (flags u8[n] with bit 8 = concurrent, updated u32[w] ascending). keysets: u32[n] key-set words (None = all K writers, in order)"""
        id, field, clocks, val = self._args(id, field, clocks, val)
        n = len(id)
        ks = None if keysets is None else np.ascontiguousarray(keysets, np.uint32)
        flags = np.zeros(max(n, 1), np.uint8); upd = np.zeros(max(n, 1), np.uint32)
        w = self._L.orc_vc_merge_batch_ks(self._h, n, _p(id, C.c_uint64), _p(field, C.c_uint32), _p(clocks, C.c_uint32),
                                          None if ks is None else _p(ks, C.c_uint32), _p(val, C.c_int64), _p(flags, C.c_uint8), _p(upd, C.c_uint32))
        return flags[:n], upd[:w].copy()

    def get_row(self, id, field, with_keyset=False):
        c = np.zeros(self.K, np.uint32); v = C.c_int64(); sp = C.c_int(); ks = np.zeros(1, np.uint32)
        ok = self._L.orc_vc_get_row_ks(self._h, int(id), int(field), _p(c, C.c_uint32), _p(ks, C.c_uint32), C.byref(v), C.byref(sp))
        if not ok:
            return None
        return (c.tolist(), v.value, bool(sp.value), int(ks[0])) if with_keyset else (c.tolist(), v.value, bool(sp.value))

    def dump_rows(self):
        n = len(self)
        id = np.zeros(n, np.uint64); field = np.zeros(n, np.uint32); clocks = np.zeros((n, self.K), np.uint32); val = np.zeros(n, np.int64)
        self._L.orc_vc_dump_rows(self._h, n, _p(id, C.c_uint64), _p(field, C.c_uint32), _p(clocks, C.c_uint32), _p(val, C.c_int64))
        return id, field, clocks, val
