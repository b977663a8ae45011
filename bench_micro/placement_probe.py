"""Is the probe kernel's time a property of WHERE the table was allocated? N tables alive at once (same rows), K1 alone on each, three passes in alternating order;
then: free the slow ones, allocate again, measure again. us per k_probe_apply launch (per-kernel HIP events, 8 launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np, torch, bmx
from bmx import synth
R, D, NB, NT = 10_000_000, 1_000_000, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
res = synth.big_resident(R, seed=1)
def to_dev(c):
    i, f, t, v = c
    return (torch.from_numpy(i.view(np.int64)).to(dev), torch.from_numpy(f.view(np.int32)).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(v).to(dev))
bs = [to_dev(synth.big_deltas(D, R, seed=2, insert_pct=10, unique=True, batch=b, drift=62500)) for b in range(3 * NB)]   # every table sees every batch ONCE: pass p applies batches [p*NB, (p+1)*NB)
applied = torch.zeros(D, dtype=torch.int32, device=dev); na = torch.zeros(1, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
def k1(e, p):
    e.profile_enable(True)
    for b in range(NB): e.merge_batch_dev(D, *bs[p * NB + b], bmx.INSERT_REFERENCE, applied=applied, n_applied=na)
    ms, n = e.profile_read(); e.profile_enable(False)
    return ms["probe_apply"] * 1e3
engines = []
for k in range(NT):
    e = bmx.Engine(22_000_000 + 40 * (D // 10)); e.load_rows(*res); engines.append(e)
    print("table %d: slots at 0x%x" % (k, e.info().table_bytes), flush=True)
for p in range(3):
    order = range(NT) if p % 2 == 0 else range(NT - 1, -1, -1)
    t = {k: k1(engines[k], p) for k in order}
    print("pass %d:" % p, " ".join("%.1f" % t[k] for k in range(NT)), flush=True)
