"""Which allocation moves the probe kernel? ONE table, and the SAME work every time: one fixed batch of 1M deltas on resident unique keys (no inserts: the table never changes), its clocks
raised by one per merge so that every delta wins every time. K1 (engine events) averaged over 10 back-to-back merges while ONE thing at a time is moved:
  A  the caller's batch columns: eight buffer sets at different addresses (pads in between)
  B  the engine's per-batch workspace: a merge of a slightly larger batch makes it reallocate (eight sizes, pads in between)
usage: python bench_micro/k1_by_allocation.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np, torch, bmx
from bmx import synth
R, D = 10_000_000, 1_000_000
dev = torch.device("cuda", 0)
e = bmx.Engine(14_000_000); rid, rf, rts, rval = synth.big_resident(R, seed=1); e.load_rows(rid, rf, rts, rval)
na = torch.zeros(1, dtype=torch.int64, device=dev)
rng = np.random.default_rng(5)
pick = rng.choice(R, D, replace=False)
bi, bf = rid[pick], rf[pick]
bv = rng.integers(0, 1 << 40, D).astype(np.int64)
clock = [int(rts.max()) + 10]
def mk_set():
    s = (torch.empty(D, dtype=torch.int64, device=dev), torch.empty(D, dtype=torch.int32, device=dev), torch.empty(D, dtype=torch.int64, device=dev), torch.empty(D, dtype=torch.int64, device=dev))
    s[0].copy_(torch.from_numpy(bi.view(np.int64))); s[1].copy_(torch.from_numpy(bf.view(np.int32))); s[3].copy_(torch.from_numpy(bv))
    return s
def k1(s, reps=10):
    for _ in range(2):
        e.sync(); clock[0] += 1; s[2].fill_(clock[0]); torch.cuda.synchronize()      # (the merge before must have read its clocks before they are raised)
        e.merge_batch_dev(D, *s, bmx.INSERT_REFERENCE, applied=None, n_applied=na)
    e.sync(); e.profile_enable(True)
    for _ in range(reps):
        e.sync(); clock[0] += 1; s[2].fill_(clock[0]); torch.cuda.synchronize()
        e.merge_batch_dev(D, *s, bmx.INSERT_REFERENCE, applied=None, n_applied=na)
    e.sync(); ms, n = e.profile_read(); e.profile_enable(False)
    assert int(na.item()) == D, int(na.item())          # every delta won
    return ms["probe_apply"] * 1e3
sets, pads = [], []
for k in range(8):
    pads.append(torch.empty((k * 61 + 23) << 20, dtype=torch.uint8, device=dev))
    sets.append(mk_set())
print("A: the caller's columns at eight places:", " ".join("%.1f" % k1(s) for s in sets), flush=True)
print("A again, same order:                    ", " ".join("%.1f" % k1(s) for s in sets), flush=True)
out = ["%.1f" % k1(sets[0])]
for k, big in enumerate(range(1_050_000, 1_450_000, 50_000)):
    pads.append(torch.empty((k * 47 + 31) << 20, dtype=torch.uint8, device=dev))
    p2 = rng.choice(R, big, replace=False); clock[0] += 1
    e.merge_batch(rid[p2], rf[p2], np.full(big, clock[0], np.int64), rng.integers(0, 1 << 40, big).astype(np.int64))
    out.append("%.1f" % k1(sets[0]))
print("B: the workspace at nine places (set 0):  ", " ".join(out), flush=True)
print("A once more (last workspace):             ", " ".join("%.1f" % k1(s) for s in sets), flush=True)
