"""K1 on SoA columns (bmx_merge_batch) vs on 32-byte records (bmx_merge_records, the exchange format), same deltas, same box, interleaved; and the
owner partition alone. Two engines with the same resident graph; config-2 batches. Per-kernel times from the engine's HIP-event profile."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bullet-js_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bmx
from bmx import synth

dev = torch.device("cuda", 0)
R, D, NB = 10_000_000, 1_000_000, 12
T0, DT = 1_000_000, 1_000_000
res = synth.big_resident(R, T0=T0, DT=DT, seed=1)
engs = [bmx.Engine(capacity_rows=22_000_000, device=0) for _ in range(2)]
for e in engs:
    e.load_rows(*res)
batches = [synth.big_deltas(D, R, seed=2, T0=T0, DT=DT, insert_pct=10, unique=True, batch=b, drift=DT // 16) for b in range(NB)]
def dv(b):
    i, f, t, v = b
    return (torch.from_numpy(i.view(np.int64)).to(dev), torch.from_numpy(f.view(np.int32)).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(v).to(dev))
dbs = [dv(b) for b in batches]
applied = torch.zeros(D + 65536, dtype=torch.int32, device=dev); n_applied = torch.zeros(1, dtype=torch.int64, device=dev)
counts = torch.zeros(16, dtype=torch.int64, device=dev)
for slack, label in ((1.0, "records, no padding"), (1.03, "records in a 1.03x slab (exchange shape, world 1)")):
    slab = int(D * slack) + (64 if slack > 1 else 0)
    recs = [torch.empty((slab, 4), dtype=torch.int64, device=dev) for _ in range(NB)]
    pe = engs[1]
    for b in range(NB):
        pe.partition_by_owner_slabs_dev(D, *dbs[b], 1, slab, recs[b], counts)
    pe.sync()
    pe.timer_start()
    for b in range(NB):
        pe.partition_by_owner_slabs_dev(D, *dbs[b], 1, slab, recs[b], counts)
    part_us = pe.timer_stop() / NB * 1e3
    if slack == 1.0:
        e = engs[0]
        e.profile_enable(True)
        for b in range(NB):
            e.merge_batch_dev(D, *dbs[b], bmx.INSERT_REFERENCE, applied=applied, n_applied=n_applied)
        ms, n = e.profile_read(); e.profile_enable(False)
        print("SoA columns:", {k: round(v * 1e3, 1) for k, v in ms.items()}, "us, %d launches" % n)
    e = bmx.Engine(capacity_rows=22_000_000, device=0); e.load_rows(*res)
    e.profile_enable(True)
    for b in range(NB):
        e.merge_records_dev(slab, recs[b], bmx.INSERT_REFERENCE, applied=applied, n_applied=n_applied)
    ms, n = e.profile_read(); e.profile_enable(False)
    print("%s:" % label, {k: round(v * 1e3, 1) for k, v in ms.items()}, "us, %d launches; owner partition alone %.1f us" % (n, part_us))
    e.close()
