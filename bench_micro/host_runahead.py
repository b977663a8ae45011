"""Can the host run ahead of the GPU? Enqueue REPS large merges (each ~90 us of GPU work) and print the host time per call
and how long the final sync takes. If the host were free to run ahead, per-call host time would stay ~10 us."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bullet-js_amd"))
import bmx
from bmx import synth
dev = torch.device("cuda", 0)
R, n = 4_000_000, 1_000_000
e = bmx.Engine(16_000_000, device=0)
e.load_rows(*synth.big_resident(R, seed=1, T0=1000, DT=1000))
d = synth.big_deltas(n, R, seed=5, T0=1000, DT=1000, insert_pct=0, hot_pct=0, hot_keys=1, unique=True, batch=0)
t = [torch.from_numpy(np.ascontiguousarray(x).view(np.int64 if x.dtype.itemsize == 8 else np.int32)).to(dev) for x in d]
applied = torch.zeros(n, dtype=torch.int32, device=dev); na = torch.zeros(1, dtype=torch.int64, device=dev)
for reps in (5, 20, 100):
    e.sync(); torch.cuda.synchronize()
    ts = []
    t0 = time.perf_counter()
    for _ in range(reps):
        a = time.perf_counter()
        e.merge_batch_dev(n, *t, bmx.INSERT_REFERENCE, applied=applied, n_applied=na)
        ts.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    e.sync()
    t2 = time.perf_counter()
    print("reps %3d: host %.1f us/call (first 5: %s), enqueue total %.0f us, then sync %.0f us" % (reps, (t1 - t0) / reps * 1e6,
          " ".join("%.0f" % (x * 1e6) for x in ts[:5]), (t1 - t0) * 1e6, (t2 - t1) * 1e6), flush=True)
