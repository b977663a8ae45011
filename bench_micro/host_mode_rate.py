"""Side measurement for DESIGN.md: merge rate when the boundary hands over HOST buffers (BMX_MEM_HOST): includes the H2D copy of
28 B/delta and the D2H copy of the winners. Synchronous calls vs the pipelined submit/collect form (upload of batch b+1 under the
merge of batch b). The C ABI is called directly with preallocated output arrays (what a native host does). Never reported as
bench.py's `value`."""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np
import bmx
from bmx import synth
R, D, NB = 10_000_000, 1_000_000, 12
res = synth.big_resident(R, seed=1)
bs = [tuple(np.ascontiguousarray(x) for x in synth.big_deltas(D, R, seed=2, insert_pct=10, unique=True, batch=b, drift=62500)) for b in range(NB)]
P = lambda a: C.c_void_p(a.ctypes.data)
applied = [np.zeros(D, np.uint32), np.zeros(D, np.uint32)]
na = C.c_uint64(); st = bmx.MergeStats()

def run(label, pipelined, reuse, pinned=False):
    global applied
    e = bmx.Engine(22_000_000); e.load_rows(*res)
    L, h = e.L, e.h
    keep = []
    if pinned:                                             # the host's two array sets and its two winner lists in page-locked memory (bmx_host_alloc)
        hc = [bmx.host_columns(D) for _ in range(2)]
        keep = [x[0] for x in hc]
        buf = [tuple(x[1:]) for x in hc]
        ho = [bmx.HostBuffer(4 * D) for _ in range(2)]; keep += ho
        applied = [o.array(np.uint32, D) for o in ho]
    else:
        buf = [tuple(np.empty_like(x) for x in bs[0]) for _ in range(2)]
        applied = [np.zeros(D, np.uint32), np.zeros(D, np.uint32)]
    def cols(b, k):
        if not reuse:
            return bs[b]
        for x, y in zip(buf[k], bs[b]): x[:] = y       # the host refills its own two sets of typed arrays (not timed below)
        return buf[k]
    e.merge_batch(*bs[0], want_flags=False)
    tot = 0.0
    if not pipelined:
        for b in range(1, NB):
            c = cols(b, b & 1)
            t0 = time.perf_counter()
            rc = L.bmx_merge_batch(h, D, P(c[0]), P(c[1]), P(c[2]), P(c[3]), 0, 0, P(applied[0]), C.cast(C.byref(na), C.c_void_p), None, C.cast(C.byref(st), C.c_void_p))
            tot += time.perf_counter() - t0
            assert rc == 0
    else:
        tk = [C.c_uint64(), C.c_uint64()]
        c = cols(1, 1)
        t0 = time.perf_counter()
        assert L.bmx_merge_submit(h, D, P(c[0]), P(c[1]), P(c[2]), P(c[3]), 0, 0, C.byref(tk[1])) == 0
        tot += time.perf_counter() - t0
        for b in range(2, NB):
            c = cols(b, b & 1)
            t0 = time.perf_counter()
            assert L.bmx_merge_submit(h, D, P(c[0]), P(c[1]), P(c[2]), P(c[3]), 0, 0, C.byref(tk[b & 1])) == 0
            assert L.bmx_merge_collect(h, tk[(b - 1) & 1], P(applied[(b - 1) & 1]), C.cast(C.byref(na), C.c_void_p), None, C.cast(C.byref(st), C.c_void_p)) == 0
            tot += time.perf_counter() - t0
        t0 = time.perf_counter()
        assert L.bmx_merge_collect(h, tk[(NB - 1) & 1], P(applied[(NB - 1) & 1]), C.cast(C.byref(na), C.c_void_p), None, C.cast(C.byref(st), C.c_void_p)) == 0
        tot += time.perf_counter() - t0
    dt = tot / (NB - 1)
    print("%-78s %5.0f us per 1M-delta batch -> %.2f G merges/s" % (label, dt * 1e6, D / dt / 1e9))
    e.close()
    for k in keep: k.close()

run("host buffers, synchronous bmx_merge_batch, fresh arrays per batch:", False, False)
run("host buffers, synchronous bmx_merge_batch, the host's own two reused array sets:", False, True)
run("host buffers, pipelined bmx_merge_submit/collect, fresh arrays per batch:", True, False)
run("host buffers, pipelined bmx_merge_submit/collect, two reused array sets:", True, True)
run("host buffers in page-locked memory (bmx_host_alloc), synchronous bmx_merge_batch:", False, True, True)
run("host buffers in page-locked memory (bmx_host_alloc), pipelined submit/collect:", True, True, True)
print("(PCIe ceiling at 52.6 GB/s for 28 B in + 3.4 B out per delta: ~1.67 G merges/s)")
