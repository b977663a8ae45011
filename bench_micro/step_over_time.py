"""Is the run-to-run spread of the step (72 ... 84 us on one box, process to process) a property of the PROCESS (where its allocations landed) or of the MOMENT (clocks, neighbours on the host)?
One process, one engine, one table: REPS groups of 20 back-to-back merges of the bench's stream (fresh batches every time), us per step of every group, with wall-clock stamps; between groups
the host generates the next batches (~1.5 s), like separate bench runs would. usage: python bench_micro/step_over_time.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np, torch, bmx
from bmx import synth
R, D, G = 10_000_000, 1_000_000, 20
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
e = bmx.Engine(R + (REPS * G + 8) * (D // 10) + (1 << 20)); e.load_rows(*synth.big_resident(R, seed=1))
print("table placement:", e.get_placement() if hasattr(e, "get_placement") else "", flush=True)
na = torch.zeros(1, dtype=torch.int64, device=dev)
def to_dev(c):
    i, f, t, v = c
    return (torch.from_numpy(i.view(np.int64)).to(dev), torch.from_numpy(f.view(np.int32)).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(v).to(dev))
t00 = time.time()
b = 0
for rep in range(REPS):
    bs = [to_dev(synth.big_deltas(D, R, seed=2, insert_pct=10, unique=True, batch=b + k, drift=62500)) for k in range(G + 3)]
    b += G + 3
    torch.cuda.synchronize()
    for k in range(3):
        e.merge_batch_dev(D, *bs[k], bmx.INSERT_REFERENCE, applied=None, n_applied=na)
    e.sync(); t0 = time.perf_counter()
    for k in range(3, G + 3):
        e.merge_batch_dev(D, *bs[k], bmx.INSERT_REFERENCE, applied=None, n_applied=na)
    e.sync(); dt = time.perf_counter() - t0
    print("group %2d at +%5.1f s: %.2f us per step (rows %d)" % (rep, time.time() - t00, dt / G * 1e6, e.row_count()), flush=True)
    del bs
