"""How long are the duplicate lists the resolve kernel walks in the streaming replay (config 5: 30 % of every 1M-delta batch on R/1000 hot keys)? CPU only
(numpy over the bench's own generator): per batch of the steady state, the deltas that beat the row they meet — the only ones that claim and link — per key.
usage: python bench_micro/config5_lists.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bullet-js_amd"))
from bmx import synth
R, D, T0, DT = 10_000_000, 1_000_000, 1_000_000, 1_000_000
ids, fld, ts, val = synth.big_resident(R, seed=1)
order = np.argsort(ids)
sid, sts, sval = ids[order], ts[order].copy(), val[order].copy()
for b in range(14):
    i, f, t, v = synth.big_deltas(D, R, seed=52, T0=T0, DT=DT, insert_pct=0, hot_pct=30, hot_keys=R // 1000, unique=False, batch=b, drift=DT // 2)
    pos = np.searchsorted(sid, i)
    gt = (t > sts[pos]) | ((t == sts[pos]) & (v > sval[pos]))          # (an upper bound: a delta that sees a value stored earlier in this batch drops out too)
    if b >= 10:
        c = np.bincount(pos[gt]); c = c[c > 1]
        print("batch %d: %d deltas beat the row they meet; %d keys with more than one: list length mean %.2f, p99 %d, max %d; %d followers"
              % (b, gt.sum(), len(c), c.mean(), np.percentile(c, 99), c.max(), (c - 1).sum()))
    o = np.lexsort((v, t, pos)); p2 = pos[o]; last = np.r_[p2[1:] != p2[:-1], True]
    bp, bt, bv = p2[last], t[o][last], v[o][last]
    better = (bt > sts[bp]) | ((bt == sts[bp]) & (bv > sval[bp]))
    sts[bp[better]] = bt[better]; sval[bp[better]] = bv[better]
