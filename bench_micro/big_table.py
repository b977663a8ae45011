"""Scale check: the merge on a resident graph of R rows (default 1e9: a 69 GB table, two hundred times the Infinity Cache), bit-exact against the CPU
oracle restricted to the rows the batches touch (decisions depend on nothing else), with per-kernel times. Not part of the test suite (minutes).
usage: python bench_micro/big_table.py [R] [batches]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bullet-js_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bmx
from bmx import synth
from oracle.oracle import Oracle

R = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 16
D = 1_000_000
T0, DT = 1_000_000, 1_000_000
dev = torch.device("cuda", 0)


def resident(rows):
    """(id, field, ts, val) of the resident rows with these ordinals: values are a function of the ordinal alone."""
    rows = np.asarray(rows, dtype=np.int64)
    ids, fld = synth.rows_to_keys(rows)
    with np.errstate(over="ignore"):
        i = rows.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        u1 = synth.splitmix64_np(i + np.uint64(0x1234567))
        u2 = synth.splitmix64_np(i + np.uint64(0x89abcdef))
    return ids, fld, (T0 + (u1 % np.uint64(DT))).astype(np.int64), (u2 % np.uint64(1 << 32)).astype(np.int64) - (1 << 31)


t0 = time.perf_counter()
e = bmx.Engine(capacity_rows=R + (NB + 2) * D, device=0, flags=bmx.CTX_FIXED_CAPACITY if hasattr(bmx, "CTX_FIXED_CAPACITY") else 0)
CH = 16_000_000
for lo in range(0, R, CH):
    e.load_rows(*resident(np.arange(lo, min(R, lo + CH), dtype=np.int64)))
    if (lo // CH) % 8 == 0:
        print("loaded %d M rows, %.0f s" % (min(R, lo + CH) // 1_000_000, time.perf_counter() - t0), flush=True)
e.sync()
info = None
print("table: %d rows resident, load took %.0f s" % (e.row_count(), time.perf_counter() - t0), info if info else "", flush=True)

batches = [synth.big_deltas(D, R, seed=5, T0=T0, DT=DT, insert_pct=10, unique=(b % 2 == 0), batch=b, drift=DT // 16) for b in range(NB)]
# oracle over the touched resident rows only
touched = []
for b in range(NB):
    j = np.arange(D, dtype=np.int64)
    if b % 2 == 0:
        touched.append(((j + b * D) * synth.PERM_PRIME + 7) % R)
    else:
        touched.append((synth._u(5 + 7919 * b, D, 4) % np.uint64(R)).astype(np.int64))
touched = np.unique(np.concatenate(touched))
o = Oracle()
o.load_rows(*resident(touched))
n_sub = len(o)


def dv(b):
    i, f, t, v = b
    return (torch.from_numpy(i.view(np.int64)).to(dev), torch.from_numpy(f.view(np.int32)).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(v).to(dev))


applied = torch.zeros(D, dtype=torch.int32, device=dev); n_applied = torch.zeros(1, dtype=torch.int64, device=dev)
e.profile_enable(True)
bad = 0
for b in range(NB):
    cols = dv(batches[b])
    e.merge_batch_dev(D, *cols, bmx.INSERT_REFERENCE, applied=applied, n_applied=n_applied)
    e.sync()
    na = int(n_applied.item())
    got = applied[:na].cpu().numpy().astype(np.uint32)
    _, want = o.merge_batch(*batches[b])
    ok = np.array_equal(got, np.asarray(want, dtype=np.uint32))
    bad += 0 if ok else 1
    print("batch %2d (%s keys): %d winners, winner list %s" % (b, "unique" if b % 2 == 0 else "random", na, "== oracle" if ok else "DIFFERS"), flush=True)
ms, n = e.profile_read(); e.profile_enable(False)
# final state of every touched or inserted row, and the row count
ids, fld, ts, val = o.dump_rows()
gts, gval, found = e.get_rows(ids, fld)
state_ok = bool(found.all()) and np.array_equal(gts, ts) and np.array_equal(gval, val)
rows_ok = e.row_count() == R + (len(o) - n_sub)
print("per-kernel us at R=%d:" % R, {k: round(v * 1e3, 1) for k, v in ms.items()}, "(%d launches)" % n)
print("final state of the %d touched/inserted rows %s; row count %d %s" % (len(ids), "== oracle" if state_ok else "DIFFERS", e.row_count(), "ok" if rows_ok else "WRONG"))
print("RESULT", "OK" if (bad == 0 and state_ok and rows_ok) else "FAILED", "total %.0f s" % (time.perf_counter() - t0))
e.close()
sys.exit(0 if (bad == 0 and state_ok and rows_ok) else 1)
