"""Side measurement for DESIGN.md: merge rate when the boundary hands over HOST buffers (BMX_MEM_HOST): includes the
H2D copy of 28 B/delta and the D2H copy of the winners. Never reported as bench.py's `value`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np
import bmx
from bmx import synth
R, D = 10_000_000, 1_000_000
e = bmx.Engine(22_000_000)
e.load_rows(*synth.big_resident(R, seed=1))
bs = [synth.big_deltas(D, R, seed=2, insert_pct=10, unique=True, batch=b, drift=62500) for b in range(8)]
e.merge_batch(*bs[0], want_flags=False)
t0 = time.perf_counter()
for b in bs[1:]:
    e.merge_batch(*b, want_flags=False)
dt = (time.perf_counter() - t0) / 7
print("host-buffer mode: %.0f us per 1M-delta batch -> %.2f G merges/s (pageable numpy buffers, synchronous call)" % (dt * 1e6, D / dt / 1e9))
e.close()
# same batches through ONE set of host buffers (what a host that reuses its typed arrays sees: the runtime has the pages pinned already)
e = bmx.Engine(22_000_000)
e.load_rows(*synth.big_resident(R, seed=1))
buf = [np.empty_like(x) for x in bs[0]]
for x, y in zip(buf, bs[0]): x[:] = y
e.merge_batch(*buf, want_flags=False)
tot = 0.0
for b in bs[1:]:
    for x, y in zip(buf, b): x[:] = y
    t0 = time.perf_counter()
    e.merge_batch(*buf, want_flags=False)
    tot += time.perf_counter() - t0
dt = tot / 7
print("host-buffer mode, reused buffers: %.0f us per 1M-delta batch -> %.2f G merges/s" % (dt * 1e6, D / dt / 1e9))
e.close()
