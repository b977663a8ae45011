"""Side experiment (not bench.py): config-5 shaped batches (30 % of deltas on R/1000 hot keys, drift DT/2) on one GPU.
Prints per-kernel times from the engine's HIP-event profile and checks the result against the oracle once."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np, torch
import bmx
from bmx import synth

R, D, NB = 10_000_000, 1_000_000, int(os.environ.get("NB", "12"))
HOT = float(os.environ.get("HOT", "30"))
dev = torch.device("cuda", 0)
def to_dev(c): return (torch.from_numpy(c[0].view(np.int64)).to(dev), torch.from_numpy(c[1].view(np.int32)).to(dev), torch.from_numpy(c[2]).to(dev), torch.from_numpy(c[3]).to(dev))
e = bmx.Engine(22_000_000)
e.load_rows(*synth.big_resident(R, seed=1))
batches = [synth.big_deltas(D, R, seed=9, insert_pct=10, hot_pct=HOT, hot_keys=R // 1000, unique=False, batch=b) for b in range(NB)]
dd = [to_dev(b) for b in batches]
applied = torch.zeros(D, dtype=torch.int32, device=dev); n_applied = torch.zeros(NB, dtype=torch.int64, device=dev); stats = torch.zeros((NB, 4), dtype=torch.int64, device=dev)
for b in range(2): e.merge_batch_dev(D, *dd[b], applied=applied, n_applied=n_applied[b:b+1], stats=stats[b])
e.sync(); e.profile_enable(True); e.timer_start()
for b in range(2, NB): e.merge_batch_dev(D, *dd[b], applied=applied, n_applied=n_applied[b:b+1], stats=stats[b])
ms = e.timer_stop(); st, n = e.profile_read(); e.profile_enable(False)
print("hot_pct", HOT, "us/step", round(ms / (NB - 2) * 1e3, 1), {k: round(v * 1e3, 1) for k, v in st.items()}, "conflicts/batch", int(stats[2:, 1].float().mean().item()), "winners/batch", int(stats[2:, 0].float().mean().item()))
if os.environ.get("CHECK", "1") == "1":
    from oracle.oracle import Oracle, rows_digest
    o = Oracle(); o.load_rows(*synth.big_resident(R, seed=1))
    for b in range(NB): o.merge_batch(*batches[b])
    print("digest equal:", rows_digest(*e.dump_rows()) == o.digest())
e.close()
