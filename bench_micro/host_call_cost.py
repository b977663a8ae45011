"""Host-side cost of the enqueue-only (BMX_MEM_DEVICE) entry points: tiny batches, so the GPU never limits."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bullet-js_amd"))
import bmx
from bmx import synth
dev = torch.device("cuda", 0)
e = bmx.Engine(1_000_000, device=0)
n = 2048
d = synth.big_deltas(n, 100000, seed=5, T0=1000, DT=1000, insert_pct=10, hot_pct=0, hot_keys=1, unique=True, batch=0)
t = [torch.from_numpy(np.ascontiguousarray(x).view(np.int64 if x.dtype.itemsize == 8 else np.int32)).to(dev) for x in d]
recs = torch.empty((n, 4), dtype=torch.int64, device=dev); counts = torch.zeros(1, dtype=torch.int64, device=dev)
applied = torch.zeros(n, dtype=torch.int32, device=dev); na = torch.zeros(1, dtype=torch.int64, device=dev)
side = torch.cuda.Stream()
def timeit(name, fn, reps=300):
    for _ in range(20): fn()
    e.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    dt = time.perf_counter() - t0
    e.sync(); torch.cuda.synchronize()
    print("%-28s %.1f us per call (host)" % (name, dt / reps * 1e6), flush=True)
timeit("partition_by_owner_slabs", lambda: e.partition_by_owner_slabs_dev(n, *t, 1, n, recs, counts))
timeit("merge_records", lambda: e.merge_records_dev(n, recs, bmx.INSERT_REFERENCE, applied=applied, n_applied=na))
timeit("merge_batch", lambda: e.merge_batch_dev(n, *t, bmx.INSERT_REFERENCE, applied=applied, n_applied=na))
timeit("order_stream_after", lambda: e.order_stream_after(side.cuda_stream))
ev = torch.cuda.Event()
timeit("torch event record+wait", lambda: (ev.record(side), torch.cuda.current_stream().wait_event(ev)))
