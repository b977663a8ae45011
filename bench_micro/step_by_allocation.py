"""Which allocation of a process decides its step time (72 ... 84 us from process to process on one box, stable inside a process: step_over_time.py)?
One engine, one (tuned) table. Groups of 20 back-to-back merges of fresh batches, us per step per group:
  phase 1  batch columns in buffer set A
  phase 2  a pad, then buffer set B (other addresses): groups alternate A / B            -> do the CALLER's buffers matter?
  phase 3  one merge of a LARGER batch makes the engine reallocate its per-batch workspace, then groups on set A again, twice    -> does the WORKSPACE matter?
usage: python bench_micro/step_by_allocation.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np, torch, bmx
from bmx import synth
R, D, G = 10_000_000, 1_000_000, 20
dev = torch.device("cuda", 0)
e = bmx.Engine(40_000_000); e.load_rows(*synth.big_resident(R, seed=1))
na = torch.zeros(1, dtype=torch.int64, device=dev)
def alloc_set(d=D):
    return [(torch.empty(d, dtype=torch.int64, device=dev), torch.empty(d, dtype=torch.int32, device=dev), torch.empty(d, dtype=torch.int64, device=dev), torch.empty(d, dtype=torch.int64, device=dev)) for _ in range(G + 3)]
def fill(bufs, b0, d=D):
    for k, s in enumerate(bufs):
        i, f, t, v = synth.big_deltas(d, R, seed=2, insert_pct=10, unique=True, batch=b0 + k, drift=62500)
        s[0].copy_(torch.from_numpy(i.view(np.int64))); s[1].copy_(torch.from_numpy(f.view(np.int32))); s[2].copy_(torch.from_numpy(t)); s[3].copy_(torch.from_numpy(v))
    torch.cuda.synchronize()
b = 0
def group(bufs, tag):
    global b
    fill(bufs, b); b += G + 3
    for k in range(3):
        e.merge_batch_dev(D, *bufs[k], bmx.INSERT_REFERENCE, applied=None, n_applied=na)
    e.sync(); t0 = time.perf_counter()
    for k in range(3, G + 3):
        e.merge_batch_dev(D, *bufs[k], bmx.INSERT_REFERENCE, applied=None, n_applied=na)
    e.sync(); dt = time.perf_counter() - t0
    print("%-34s %.2f us per step (rows %d)" % (tag, dt / G * 1e6, e.row_count()), flush=True)
A = alloc_set()
print("set A: ids of its first batch at 0x%x" % A[0][0].data_ptr())
for _ in range(3): group(A, "phase 1, set A")
pad = torch.empty(1500 << 20, dtype=torch.uint8, device=dev)
Bs = alloc_set()
print("set B: ids of its first batch at 0x%x" % Bs[0][0].data_ptr())
for _ in range(3):
    group(Bs, "phase 2, set B"); group(A, "phase 2, set A")
for big in (1_300_000, 1_700_000):
    i, f, t, v = synth.big_deltas(big, R, seed=9, insert_pct=10, unique=True, batch=1000 + big, drift=62500)
    e.merge_batch(i, f, t, v)          # a larger batch: the per-batch workspace grows, i.e. moves
    for _ in range(3): group(A, "phase 3, workspace for %d, set A" % big)
