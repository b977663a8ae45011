"""Scale check of the index paths: R rows (default 1e9) of ONE integer field, values in [0, 1000); the column scan (bmx_scan_range / _count) and the
value-ordered view (bmx_index_set_ordered) answer the same queries, every answer checked against numpy (match count and wrap-around sum of the ids,
accumulated while the rows are generated). Not part of the test suite (a minute, ~125 GB of HBM). usage: python bench_micro/big_index.py [R]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bullet-js_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bmx
from bmx import synth

R = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
dev = torch.device("cuda", 0)
F = synth.fnv1a32("n:age")
QUERIES = [("equals 0.1 %", 42, 42), ("range 1 %", 100, 109), ("range 10 %", 100, 199)]
want = {q[0]: [0, 0] for q in QUERIES}
t0 = time.perf_counter()
e = bmx.Engine(capacity_rows=R + 1024, device=0)
CH = 16_000_000
for lo in range(0, R, CH):
    m = min(CH, R - lo)
    ids = synth.splitmix64_np(np.arange(lo + 1, lo + m + 1, dtype=np.uint64))
    with np.errstate(over="ignore"):
        ages = (synth.splitmix64_np(ids ^ np.uint64(0xABCDEF)) % np.uint64(1000)).astype(np.int64)
        for name, a, b in QUERIES:
            sel = (ages >= a) & (ages <= b)
            want[name][0] += int(sel.sum()); want[name][1] = (want[name][1] + int(ids[sel].sum(dtype=np.uint64))) & ((1 << 64) - 1)
    e.load_rows(ids, np.full(m, F, np.uint32), np.full(m, 5, np.int64), ages)
    if (lo // CH) % 16 == 0:
        print("loaded %d M rows, %.0f s" % ((lo + m) // 1_000_000, time.perf_counter() - t0), flush=True)
e.sync()
print("table: %d rows resident (%.1f GB), load took %.0f s" % (e.row_count(), e.info().table_bytes / 1e9, time.perf_counter() - t0), flush=True)
t1 = time.perf_counter(); e.index_build(F); e.sync()
print("index built in %.1f ms (%d positions)" % ((time.perf_counter() - t1) * 1e3, e.index_size(F)), flush=True)
cap = max(w[0] for w in want.values()) + 16
out = torch.zeros(cap, dtype=torch.int64, device=dev); n_out = torch.zeros(1, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
i64 = lambda x: x - (1 << 64) if x >= (1 << 63) else x


def run(label):
    ok = True
    for name, a, b in QUERIES:
        for _ in range(2):
            e.scan_range_dev(F, a, b, out, cap, n_out)
        e.sync(); e.timer_start()
        for _ in range(5):
            e.scan_range_dev(F, a, b, out, cap, n_out)
        us = e.timer_stop() / 5 * 1e3
        m = int(n_out.item())
        good = m == want[name][0] and int(out[:m].sum().item()) == i64(want[name][1])
        e.sync(); e.timer_start()
        for _ in range(5):
            e.scan_range_dev(F, a, b, None, 0, n_out)
        us_cnt = e.timer_stop() / 5 * 1e3
        good = good and int(n_out.item()) == want[name][0]
        ok = ok and good
        print("%-22s %-13s %10d matches: %9.1f us with ids, %8.1f us count only  %s" % (label, name, m, us, us_cnt, "== numpy" if good else "DIFFERS"), flush=True)
    return ok


ok = run("column scan")
e.index_set_ordered(F, 1)
e.sync(); t1 = time.perf_counter()
e.scan_range_dev(F, 0, 0, None, 0, n_out); e.sync()
print("value-ordered view sorted in %.1f ms; valid: %s" % ((time.perf_counter() - t1) * 1e3, e.index_ordered_info(F)[1]), flush=True)
ok = run("value-ordered view") and ok and e.index_ordered_info(F)[1]
print("RESULT %s total %.0f s" % ("OK" if ok else "FAILED", time.perf_counter() - t0))
e.close()
sys.exit(0 if ok else 1)
