"""Where a 1M-delta HOST batch (bmx_merge_batch, BMX_MEM_HOST) spends its time: pageable caller arrays (fresh every call, and one set reused)
against page-locked ones from bmx_host_alloc (inputs only, inputs and the winner list). Outputs are preallocated and touched, so the Python
wrapper's own allocations are not in the figure."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np
import bmx
from bmx import synth
R, D, NB = 10_000_000, 1_000_000, 9
g = bmx.Engine(22_000_000)
g.load_rows(*synth.big_resident(R, seed=1))
L, P = g.L, bmx._ptr
KINDS = ("pageable_fresh", "pageable_reused", "pinned_in", "pinned_in_out")

def run(kind):
    ts = []
    applied = np.ones(D, np.uint32); na = C.c_uint64(0); st = bmx.MergeStats()
    hb = hbo = None
    if kind.startswith("pinned"):
        hb, *pin = bmx.host_columns(D)
    if kind == "pinned_in_out":
        hbo = bmx.HostBuffer(4 * D); applied = hbo.array(np.uint32, D)
    reuse = None
    for b in range(NB):
        cols = synth.big_deltas(D, R, seed=7, insert_pct=10, unique=True, batch=40 + b + 100 * KINDS.index(kind), drift=62500)
        if hb:
            for d, s in zip(pin, cols): d[:] = s
            cols = pin
        elif kind == "pageable_reused":
            if reuse is None: reuse = [c.copy() for c in cols]
            for d, s in zip(reuse, cols): d[:] = s
            cols = reuse
        t0 = time.perf_counter()
        rc = L.bmx_merge_batch(g.h, D, P(cols[0]), P(cols[1]), P(cols[2]), P(cols[3]), bmx.INSERT_REFERENCE, bmx.MEM_HOST, P(applied),
                               C.cast(C.byref(na), C.c_void_p), None, C.cast(C.byref(st), C.c_void_p))
        ts.append(time.perf_counter() - t0)
        assert rc == 0 and na.value > 0
    ts = sorted(ts[1:])
    print("%-15s 1M-delta host batch: median %.0f us, best %.0f us (%d winners back)" % (kind, ts[len(ts) // 2] * 1e6, ts[0] * 1e6, na.value), flush=True)

for kind in KINDS:
    run(kind)
