"""GPU parity, randomized: many small shapes against the oracle (merge, scans) and against a numpy stable sort (owner partition).
Sizes sit on the kernels' internal boundaries (64-lane waves, 256-delta blocks, 1024-delta partition tiles, 4096-delta compaction
blocks); timestamps and values come from tiny ranges so ties and duplicate keys are the norm, plus the domain's extreme values."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import bmx
from oracle import streams
from oracle.oracle import Oracle, INSERT_REFERENCE, INSERT_DELTA, rows_digest, owner_of as o_owner

VMAX = 2**53 - 1


def _batch(rng, n, nkeys, nfields, tmax, vr, extremes):
    rows = rng.integers(0, max(nkeys, 1), n)
    ids = streams.splitmix64_np(rows.astype(np.uint64) + np.uint64(1)) if n else np.zeros(0, np.uint64)
    ftab = np.array([streams.field_hash(x) for x in range(nfields)], np.uint32)
    fields = ftab[rng.integers(0, nfields, n)]
    ts = rng.integers(0, tmax + 1, n).astype(np.int64)
    val = rng.integers(-vr, vr + 1, n).astype(np.int64)
    if extremes and n:
        k = rng.integers(0, n, max(1, n // 10))
        ts[k] = rng.choice(np.array([0, 1, 2, 3, 2**31 - 1, 2**31, 2**31 + 1, VMAX], np.int64), len(k))
        k = rng.integers(0, n, max(1, n // 10))
        val[k] = rng.choice(np.array([-VMAX, -1, 0, 1, VMAX], np.int64), len(k))
    return ids, fields, ts, val


SIZES = [0, 1, 2, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 9000]
# soak mode (spare GPU time): BMX_FUZZ_SEEDS=<count> [BMX_FUZZ_SEED0=<first>] runs more seeds, BMX_FUZZ_BIG=1 adds batches that span many
# compute units at once (races between waves of different CUs on one key or one line)
N_SEEDS = int(os.environ.get("BMX_FUZZ_SEEDS", "12"))
SEED0 = int(os.environ.get("BMX_FUZZ_SEED0", "0"))
BIG = os.environ.get("BMX_FUZZ_BIG") == "1"
if BIG:
    SIZES = SIZES + [65536, 200000, 300001]


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + N_SEEDS))
def test_merge_random_shapes_match_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    for case in range(25):
        mode = [INSERT_REFERENCE, INSERT_DELTA][int(rng.integers(0, 2))]
        strict = bool(rng.integers(0, 2))
        nkeys = int(rng.choice([1, 3, 50, 2000, 100000] + ([20000, 1000000] if BIG else [])))
        nfields = int(rng.choice([1, 2, 5]))
        tmax = int(rng.choice([1, 4, 1000]))
        vr = int(rng.choice([0, 1, 100]))
        e = bmx.Engine(int(rng.choice([64, 5000, 50000]))); o = Oracle()
        nres = int(rng.choice(SIZES))
        res = _batch(rng, nres, nkeys, nfields, tmax, vr, extremes=bool(rng.integers(0, 2)))
        if nres:                                   # a preload holds every key once
            _, first = np.unique(np.stack([res[0], res[1].astype(np.uint64)], axis=1), axis=0, return_index=True)
            res = tuple(x[np.sort(first)] for x in res)
        e.load_rows(*res); o.load_rows(*res)
        for b in range(int(rng.integers(1, 4))):
            n = int(rng.choice(SIZES))
            d = _batch(rng, n, nkeys, nfields, tmax + b, vr, extremes=bool(rng.integers(0, 2)))
            applied, flags, st = e.merge_batch(*d, insert_mode=mode | (bmx.MERGE_STRICT_FLAGS if strict else 0), want_flags=True)
            of, ow = o.merge_batch(*d, mode)
            assert np.array_equal(applied, ow), (seed, case, b, n)
            if strict:
                assert np.array_equal(flags, of), (seed, case, b, n, np.nonzero(flags != of)[0][:8])
            if n:
                assert st.n_rows == len(o)
        assert rows_digest(*e.dump_rows()) == o.digest(), (seed, case)
        e.close(); o.close()


@pytest.mark.parametrize("seed", range(3))
def test_scans_random_columns_match_oracle(seed):
    rng = np.random.default_rng(2000 + seed)
    for case in range(6):
        n = int(rng.choice([1, 100, 8191, 8192, 8193, 70000]))
        wide = bool(rng.integers(0, 2))
        span = 2**40 if wide else int(rng.choice([3, 100, 100000]))
        ids = np.array([streams.splitmix64(i + 1) for i in range(n)], np.uint64)
        fa, fb = streams.field_hash(1), streams.field_hash(2)
        va = rng.integers(-span, span + 1, n).astype(np.int64); vb = rng.integers(0, 10, n).astype(np.int64)
        e = bmx.Engine(max(2 * n, 64)); o = Oracle()
        for f, v in ((fa, va), (fb, vb)):
            e.load_rows(ids, np.full(n, f, np.uint32), np.full(n, 5, np.int64), v)
            o.load_rows(ids, np.full(n, f, np.uint32), np.full(n, 5, np.int64), v)
        e.index_build(fa); e.index_build(fb)
        for q in range(8):
            lo = int(rng.integers(-span - 2, span + 3)); hi = lo + int(rng.integers(-2, span + 3))
            got = np.sort(e.scan_range(fa, lo, hi)); ref = np.sort(o.scan_range(fa, lo, hi))
            assert np.array_equal(got, ref), (seed, case, q, lo, hi)
            assert e.scan_count(fa, lo, hi) == len(ref)
            t2 = (int(rng.integers(0, 10)), int(rng.integers(0, 12)))
            got = np.sort(e.scan_filter([(fa, lo, hi), (fb, min(t2), max(t2))]))
            ref = np.sort(o.scan_filter_and([(fa, lo, hi), (fb, min(t2), max(t2))]))
            assert np.array_equal(got, ref), (seed, case, q)
        e.close(); o.close()


@pytest.mark.parametrize("nshards", [1, 2, 3, 5, 8, 13, 16])
def test_partition_random_sizes_are_stable(nshards):
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3000 + nshards)
    with bmx.Engine(1024) as e:
        for n in [1, 63, 64, 65, 1023, 1024, 1025, 2049, 5000, 300001]:
            skew = bool(rng.integers(0, 2))
            ids, fields, ts, val = _batch(rng, n, 7 if skew else 10**6, 2, 1000, 100, extremes=False)   # skew: few distinct owners
            dd = [torch.from_numpy(np.ascontiguousarray(x).view(np.int64 if x.dtype.itemsize == 8 else np.int32)).to(dev) for x in (ids, fields, ts, val)]
            own = o_owner(ids, nshards)
            want_counts = np.bincount(own, minlength=nshards)
            order = np.argsort(own, kind="stable")
            # compact form
            recs = torch.zeros((n, 4), dtype=torch.int64, device=dev); counts = torch.zeros(nshards, dtype=torch.int64, device=dev)
            e.partition_by_owner_dev(n, *dd, nshards, recs, counts); e.sync()
            r = recs.cpu().numpy().view(bmx.DELTA_REC_DTYPE).reshape(-1)
            assert counts.cpu().tolist() == want_counts.tolist(), (n, nshards)
            assert np.array_equal(r["aux"], order.astype(np.uint32)), (n, nshards)
            assert np.array_equal(r["id"], ids[order]) and np.array_equal(r["ts"], ts[order]) and np.array_equal(r["val"], val[order])
            assert np.array_equal(r["field"], fields[order])
            # slab form, slabs deliberately too small for the fullest shard when skewed: overflow drops records, counts still tell
            slab = int(want_counts.max()) if not skew else max(1, int(want_counts.max()) - 3)
            recs2 = torch.zeros((nshards * slab, 4), dtype=torch.int64, device=dev)
            e.partition_by_owner_slabs_dev(n, *dd, nshards, slab, recs2, counts)
            if int(want_counts.max()) > slab:      # dropped records are a sticky error of the context, not only a count to look at
                with pytest.raises(bmx.BmxError) as ei:
                    e.sync()
                assert ei.value.code == bmx.ERR_OVERFLOW
            else:
                e.sync()
            assert counts.cpu().tolist() == want_counts.tolist()
            r2 = recs2.cpu().numpy().view(bmx.DELTA_REC_DTYPE).reshape(nshards, slab)
            off = 0
            for g in range(nshards):
                k = min(int(want_counts[g]), slab)
                src = order[off:off + k]
                assert np.array_equal(r2[g, :k]["aux"], src.astype(np.uint32)), (n, nshards, g)
                assert np.array_equal(r2[g, :k]["id"], ids[src])
                assert (r2[g, k:]["id"] == np.uint64(2**64 - 1)).all()          # padding
                off += int(want_counts[g])
