"""Owner-partition rate (K7) for 1..8 shards on one GPU: the kernels the N>1 step runs before its all-to-all.
Run under rocprofv3 --kernel-trace --stats for per-kernel times; prints HIP-event times per call."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bullet-js_amd"))
import bmx
from bmx import synth

D = 1_000_000
dev = torch.device("cuda", 0)
e = bmx.Engine(1024, device=0)
d = synth.big_deltas(D, 10_000_000, seed=5, T0=1_000_000, DT=1_000_000, insert_pct=10, hot_pct=0, hot_keys=1, unique=True, batch=0)
t = [torch.from_numpy(np.ascontiguousarray(x).view(np.int64 if x.dtype.itemsize == 8 else np.int32)).to(dev) for x in d]
for W in (1, 2, 4, 8):
    slab = int(D / W * 1.03) + 64
    recs = torch.empty((W * slab, 4), dtype=torch.int64, device=dev)
    counts = torch.zeros(W, dtype=torch.int64, device=dev)
    for it in range(3):
        e.partition_by_owner_slabs_dev(D, *t, W, slab, recs, counts)
    e.sync()
    e.timer_start()
    for it in range(20):
        e.partition_by_owner_slabs_dev(D, *t, W, slab, recs, counts)
    ms = e.timer_stop()
    print("shards %d: %.1f us per partition of %d deltas; counts %s" % (W, ms * 1000 / 20, D, counts.tolist()), flush=True)
