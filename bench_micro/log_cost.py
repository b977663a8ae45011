"""What the index change log costs the merge: config-2 batches with and without a maintained index on the merged field (per-kernel HIP-event times),
and what the scan that follows pays to apply the log."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bullet-js_amd"))
import numpy as np, torch
import bmx
from bmx import synth

dev = torch.device("cuda", 0)
R, D, NB = 10_000_000, 1_000_000, 12
res = synth.big_resident(R)
f0 = int(res[1][0])
def dv(b):
    i, f, t, v = b
    return (torch.from_numpy(i.view(np.int64)).to(dev), torch.from_numpy(f.view(np.int32)).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(v).to(dev))
dbs = [dv(synth.big_deltas(D, R, seed=2, insert_pct=10, unique=True, batch=b, drift=1_000_000 // 16)) for b in range(2 * NB)]
applied = torch.zeros(D, dtype=torch.int32, device=dev); na = torch.zeros(1, dtype=torch.int64, device=dev)
out_ids = torch.zeros(R + 4 * D, dtype=torch.int64, device=dev)
for with_index in (False, True):
    e = bmx.Engine(capacity_rows=22_000_000, device=0); e.load_rows(*res)
    if with_index:
        e.index_build(f0)
    e.profile_enable(True)
    for b in range(NB):
        e.merge_batch_dev(D, *dbs[b], bmx.INSERT_REFERENCE, applied=applied, n_applied=na)
    ms, n = e.profile_read(); e.profile_enable(False)
    print("index maintained: %s |" % with_index, {k: round(v * 1e3, 1) for k, v in ms.items()}, "us per batch")
    if with_index:
        for b in range(NB, NB + 4):
            e.merge_batch_dev(D, *dbs[b], bmx.INSERT_REFERENCE, applied=applied, n_applied=na)
            e.sync(); t0 = time.perf_counter()
            e.scan_range_dev(f0, 0, 1000, out_ids, R, na); e.sync()
            print("  first scan after one more batch: %.0f us" % ((time.perf_counter() - t0) * 1e6), e.index_refresh_counts())
    e.close()
