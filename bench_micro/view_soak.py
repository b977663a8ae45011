"""Soak of the value-ordered view under writes: hundreds of merge -> query rounds on one index, every answer held against a numpy model of the rows.
Batch sizes from a handful to a tenth of the index (so the change run goes through every path: pending patch, cancellation of pending inserts, direct rewrite,
background rewrite), updates / re-updates / new rows / tombstones / revived rows, narrow and wide value domains (many ties on the value: the position decides),
queries in host mode and as positions. usage: python bench_micro/view_soak.py [rows] [rounds] [int32|wide] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np, bmx
from bmx import synth
VAL_DELETED = -(1 << 63)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
sh = 33 if wide else 0
rng = np.random.default_rng(seed)
F = 777
DOM = int(rng.choice([7, 300, 100000]))                      # distinct values: few (long tie runs), some, many
ids = synth.splitmix64_np(np.arange(1, R + 1, dtype=np.uint64))
vals = rng.integers(0, DOM, R).astype(np.int64) << sh
alive = np.ones(R, bool)
clock = 10
t_start = time.time()
with bmx.Engine(capacity_rows=4 * R, device=0) as e:
    e.load_rows(ids, np.full(R, F, np.uint32), np.full(R, clock, np.int64), vals)
    e.index_build(F); e.index_set_ordered(F, 1)
    e.scan_count(F, 0, 1)
    next_new = 1 << 40
    for rnd in range(ROUNDS):
        n = len(ids)
        kind = rng.integers(0, 10)
        m = int(rng.choice([3, 200, 5000, 60000, n // 10]))
        clock += 1
        if kind < 6:                                           # updates (some rows twice in the batch: equal clocks, the larger value wins)
            k = rng.choice(n, m, replace=True)
            nv = rng.integers(0, DOM, m).astype(np.int64) << sh
            e.merge_batch(ids[k], np.full(m, F, np.uint32), np.full(m, clock, np.int64), nv)
            order = np.lexsort((nv, k)); ks, vs = k[order], nv[order]
            last = np.r_[ks[1:] != ks[:-1], True]
            vals[ks[last]] = vs[last]; alive[ks[last]] = True
        elif kind < 8:                                         # new rows (some below / above every value)
            m = max(1, m // 4)
            nid = synth.splitmix64_np(np.arange(next_new, next_new + m, dtype=np.uint64)); next_new += m
            nv = rng.integers(-3, DOM + 3, m).astype(np.int64) << sh
            if len(ids) + m > 3 * R: continue
            e.merge_batch(nid, np.full(m, F, np.uint32), np.full(m, 5, np.int64), nv)
            ids = np.concatenate([ids, nid]); vals = np.concatenate([vals, nv]); alive = np.concatenate([alive, np.ones(m, bool)])
        else:                                                  # tombstones
            k = np.unique(rng.choice(n, max(1, m // 3), replace=True))
            e.put_rows(ids[k], np.full(len(k), F, np.uint32), np.full(len(k), clock, np.int64), np.full(len(k), VAL_DELETED, np.int64))
            alive[k] = False
        if rng.integers(0, 4) == 0:
            continue                                           # several writes between two queries
        for _ in range(int(rng.integers(1, 4))):
            a = int(rng.integers(-2, DOM + 2)); b = a + int(rng.choice([0, 0, 1, DOM // 3 + 1]))
            lo, hi = a << sh, b << sh
            want = np.sort(ids[(vals >= lo) & (vals <= hi) & alive])
            got = e.scan_range(F, lo, hi)
            if len(got) != len(want) or not np.array_equal(np.sort(got), want):
                raise SystemExit("MISMATCH round %d range [%d, %d]: %d rows, numpy %d; stats %s" % (rnd, a, b, len(got), len(want), e.index_ordered_stats(F)))
            if e.scan_count(F, lo, hi) != len(want):
                raise SystemExit("COUNT MISMATCH round %d" % rnd)
        if rnd % 10 == 0:
            pos = e.scan_range_pos(F, -(1 << 62), 1 << 62); col = e.index_ids(F)
            if len(pos) != int(alive.sum()) or not np.array_equal(np.sort(col[pos]), np.sort(ids[alive])):
                raise SystemExit("WHOLE-VIEW MISMATCH round %d" % rnd)
        if rnd % 25 == 0:
            print("round %d ok: %d rows, %s, %.0f s" % (rnd, len(ids), e.index_ordered_stats(F), time.time() - t_start), flush=True)
    st = e.index_ordered_stats(F)
    print("index refreshes (full rebuilds, from the change log):", e.index_refresh_counts())
    print("SOAK OK: %d rounds, %d rows at the end, domain %d%s, seed %d: %s" % (ROUNDS, len(ids), DOM, " (wide)" if wide else "", seed, st))
    assert st["sorts"] <= 2, st
