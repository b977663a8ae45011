"""Latency of small host batches (the reference's sync chunks are 50 entries): bmx_merge_batch(BMX_MEM_HOST) per call, by batch size."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bullet-js_amd"))
import numpy as np
import bmx
from bmx import synth

R = 1_000_000
e = bmx.Engine(capacity_rows=4_000_000, device=0)
e.load_rows(*synth.big_resident(R))
rng = np.random.default_rng(1)
for n in (1, 50, 1000, 10_000, 30_000, 100_000):
    batches = []
    for b in range(60):
        rows = rng.integers(0, R, n)
        ids, fld = synth.rows_to_keys(rows)
        batches.append((ids, fld, rng.integers(1_000_000, 3_000_000, n).astype(np.int64), rng.integers(-1000, 1000, n).astype(np.int64)))
    for b in batches[:10]:
        e.merge_batch(*b)
    t0 = time.perf_counter()
    for b in batches[10:]:
        e.merge_batch(*b)
    dt = (time.perf_counter() - t0) / 50
    print("n = %6d: %7.1f us per call  (%.2f M deltas/s)" % (n, dt * 1e6, n / dt / 1e6))
e.close()
