"""Does the probe kernel's time depend on WHERE THE BATCH COLUMNS live (not the table)? One engine, one table; several sets of column buffers allocated at different
times (pads in between shift them); every step uploads a FRESH batch of the bench's stream into one set (arms alternate) and merges it; K1 time per arm from the
engine's own events. Arms that differ consistently would mean the caller's buffers are part of the step's luck of the draw."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np, torch, bmx
from bmx import synth
R, D = 10_000_000, 1_000_000
ARMS, STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 48
dev = torch.device("cuda", 0)
res = synth.big_resident(R, seed=1)
e = bmx.Engine(22_000_000); e.load_rows(*res)
sets, pads = [], []
for k in range(ARMS):
    pads.append(torch.empty((k * 53 + 17) << 20, dtype=torch.uint8, device=dev))
    sets.append((torch.empty(D, dtype=torch.int64, device=dev), torch.empty(D, dtype=torch.int32, device=dev), torch.empty(D, dtype=torch.int64, device=dev), torch.empty(D, dtype=torch.int64, device=dev)))
applied = torch.zeros(D, dtype=torch.int32, device=dev); na = torch.zeros(1, dtype=torch.int64, device=dev); st = torch.zeros(4, dtype=torch.int64, device=dev)
print("placement of the table:", e.get_placement() if hasattr(e, "get_placement") else "?")
for k, s in enumerate(sets):
    print("arm %d: ids at 0x%x" % (k, s[0].data_ptr()))
per = [[] for _ in range(ARMS)]
for step in range(STEPS):
    arm = step % ARMS
    i, f, t, v = synth.big_deltas(D, R, seed=2, insert_pct=10, unique=True, batch=step, drift=62500)
    sets[arm][0].copy_(torch.from_numpy(i.view(np.int64))); sets[arm][1].copy_(torch.from_numpy(f.view(np.int32))); sets[arm][2].copy_(torch.from_numpy(t)); sets[arm][3].copy_(torch.from_numpy(v))
    torch.cuda.synchronize()
    e.profile_enable(True)
    e.merge_batch_dev(D, *sets[arm], bmx.INSERT_REFERENCE, applied=applied, n_applied=na, stats=st)
    e.sync()
    ms, n = e.profile_read(); e.profile_enable(False)
    if step >= ARMS:
        per[arm].append(ms["probe_apply"] * 1e3)
for k in range(ARMS):
    a = np.array(per[k])
    print("arm %d: probe_apply %.1f us mean, %.1f min, %.1f max over %d merges" % (k, a.mean(), a.min(), a.max(), len(a)))
