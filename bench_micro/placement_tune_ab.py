"""Does the placement tuning of bmx_create pick tables on which the REAL merge kernel is fast? Engines created with BMX_TABLE_PLACEMENT_TRIES = 1 (first allocation) and 4
(tuned), alternating, all alive at once; then k_probe_apply alone on each (8 launches of fresh batches per pass, two passes). us per launch + what the tuner saw."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np, torch, bmx
from bmx import synth
R, D, NB, NT = 10_000_000, 1_000_000, 8, 6
dev = torch.device("cuda", 0)
res = synth.big_resident(R, seed=1)
def to_dev(c):
    i, f, t, v = c
    return (torch.from_numpy(i.view(np.int64)).to(dev), torch.from_numpy(f.view(np.int32)).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(v).to(dev))
bs = [to_dev(synth.big_deltas(D, R, seed=2, insert_pct=10, unique=True, batch=b, drift=62500)) for b in range(2 * NB)]
applied = torch.zeros(D, dtype=torch.int32, device=dev); na = torch.zeros(1, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
engines = []
for k in range(NT):
    os.environ["BMX_TABLE_PLACEMENT_TRIES"] = "4" if k % 2 else "1"
    e = bmx.Engine(22_000_000 + 40 * (D // 10)); e.load_rows(*res); engines.append(e)
    print("table %d (%s): %s" % (k, "tuned" if k % 2 else "first allocation", e.placement()), flush=True)
for p in range(2):
    out = []
    for e in engines:
        e.profile_enable(True)
        for b in range(NB): e.merge_batch_dev(D, *bs[p * NB + b], bmx.INSERT_REFERENCE, applied=applied, n_applied=na)
        ms, n = e.profile_read(); e.profile_enable(False)
        out.append(ms["probe_apply"] * 1e3)
    print("pass %d: k_probe_apply us per launch:" % p, " ".join("%.1f" % x for x in out), "| first-allocation tables %.1f, tuned tables %.1f (means)" % (np.mean(out[0::2]), np.mean(out[1::2])), flush=True)
