"""Does the placement of the engine's per-batch WORKSPACE move the probe kernel? One engine, one table, batches WITHOUT inserts (the table stays as it is), groups of 20
back-to-back merges; between groups the workspace is made to move (one merge of a larger batch reallocates it). K1 from the engine's events, us per launch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np, torch, bmx
from bmx import synth
R, D, G = 10_000_000, 1_000_000, 20
dev = torch.device("cuda", 0)
na = torch.zeros(1, dtype=torch.int64, device=dev)
bufs = [(torch.empty(D, dtype=torch.int64, device=dev), torch.empty(D, dtype=torch.int32, device=dev), torch.empty(D, dtype=torch.int64, device=dev), torch.empty(D, dtype=torch.int64, device=dev)) for _ in range(G)]
b = 0
def fill():
    global b
    for s in bufs:
        i, f, t, v = synth.big_deltas(D, R, seed=2, insert_pct=0, unique=True, batch=b, drift=62500); b += 1
        s[0].copy_(torch.from_numpy(i.view(np.int64))); s[1].copy_(torch.from_numpy(f.view(np.int32))); s[2].copy_(torch.from_numpy(t + b)); s[3].copy_(torch.from_numpy(v))
    torch.cuda.synchronize()
def group(e, tag):
    fill()
    e.profile_enable(True)
    for s in bufs[:12]:
        e.merge_batch_dev(D, *s, bmx.INSERT_REFERENCE, applied=None, n_applied=na)
    e.sync()
    ms, n = e.profile_read(); e.profile_enable(False)
    e.sync(); t0 = time.perf_counter()
    for s in bufs[12:]:
        e.merge_batch_dev(D, *s, bmx.INSERT_REFERENCE, applied=None, n_applied=na)
    e.sync(); dt = (time.perf_counter() - t0) / (G - 12) * 1e6
    print("%-44s K1 %.2f us  (stream of 8: %.2f us per step)" % (tag, ms["probe_apply"] * 1e3, dt), flush=True)
for inc in range(3):
    e = bmx.Engine(14_000_000); e.load_rows(*synth.big_resident(R, seed=1))
    print("engine %d: table placement %s" % (inc, e.get_placement() if hasattr(e, "get_placement") else ""))
    group(e, "engine %d, first workspace" % inc); group(e, "engine %d, first workspace (again)" % inc)
    for big in (1_200_000, 1_500_000, 1_900_000):
        i, f, t, v = synth.big_deltas(big, R, seed=9, insert_pct=0, unique=True, batch=5000 + big + inc, drift=62500)
        e.merge_batch(i, f, t + 100000, v)
        group(e, "engine %d, workspace for %d deltas" % (inc, big))
    e.close()
