"""Per-call latency of the small host-mode queries the JS host issues: scans over small indexes, point reads."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bullet-js_amd"))
import numpy as np
import bmx
from bmx import synth

def bench(f, reps=200):
    for _ in range(20): f()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e6

for R in (10_000, 1_000_000):
    e = bmx.Engine(capacity_rows=4 * R, device=0)
    ids = synth.splitmix64_np(np.arange(1, R + 1, dtype=np.uint64))
    fa = synth.fnv1a32("n:age")
    e.load_rows(ids, np.full(R, fa, np.uint32), np.full(R, 5, np.int64), (ids % np.uint64(1000)).astype(np.int64))
    e.index_build(fa)
    print("R = %8d | scan_range equals (host ids out): %6.1f us | scan_count: %6.1f us | get_rows(1): %6.1f us | get_rows(100): %6.1f us | row_count: %5.1f us" % (
        R, bench(lambda: e.scan_range(fa, 42, 42)), bench(lambda: e.scan_count(fa, 42, 42)), bench(lambda: e.get_rows(ids[:1], np.full(1, fa, np.uint32))),
        bench(lambda: e.get_rows(ids[:100], np.full(100, fa, np.uint32))), bench(lambda: e.row_count())))
    e.close()
