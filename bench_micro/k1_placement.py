"""Does the probe kernel's time depend on WHERE the table was allocated? Several tables in one process (same rows, same batches), kernel time per table."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np, torch, bmx
from bmx import synth
R, D, NB = 10_000_000, 1_000_000, 12
dev = torch.device("cuda", 0)
res = synth.big_resident(R, seed=1)
def to_dev(c):
    i, f, t, v = c
    return (torch.from_numpy(i.view(np.int64)).to(dev), torch.from_numpy(f.view(np.int32)).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(v).to(dev))
bs = [to_dev(synth.big_deltas(D, R, seed=2, insert_pct=10, unique=True, batch=b, drift=62500)) for b in range(NB)]
applied = torch.zeros(D, dtype=torch.int32, device=dev); na = torch.zeros(1, dtype=torch.int64, device=dev); st = torch.zeros(4, dtype=torch.int64, device=dev)
engines, pads = [], []
kinds = []
for k in range(6):
    if k % 2: pads.append(torch.empty((k * 37 + 11) << 20, dtype=torch.uint8, device=dev))     # shift the next allocation
    kinds.append("table%d" % k)
    e = bmx.Engine(22_000_000); e.load_rows(*res); engines.append(e)
print("tables:", " ".join(kinds))
for rnd in range(1):
    out = []
    for e in engines:
        e.profile_enable(True)
        for b in range(NB):
            e.merge_batch_dev(D, *bs[b], bmx.INSERT_REFERENCE, applied=applied, n_applied=na, stats=st)
        e.sync()
        ms, n = e.profile_read(); e.profile_enable(False)
        out.append(ms["probe_apply"] * 1e3)
    print("round %d: probe_apply us per table:" % rnd, " ".join("%.1f" % x for x in out))
