"""GPU parity: index build + range/equals/count/filter scans (through the C ABI) vs the CPU oracle and the
reference's own query results (tests/golden/g5_*.json)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import bmx
from oracle import streams
from oracle.oracle import Oracle, INSERT_REFERENCE
from helpers import load_golden

FA, FS = streams.fnv1a32("age"), streams.fnv1a32("score")


def _nodes(N, seed):
    rng = streams.XorShift32(seed)
    ages = np.zeros(N, np.int64); scores = np.zeros(N, np.int64)
    for i in range(N):
        ages[i] = rng() % 100
        scores[i] = rng() % 200001 - 100000
    ids = np.array([streams.fnv1a32("n/k%d" % i) | (i << 32) for i in range(N)], dtype=np.uint64)
    ts = np.array([10 + (i % 7) for i in range(N)], np.int64)
    return ids, ts, ages, scores


def _load(e, ids, ts, ages, scores):
    N = len(ids)
    e.merge_batch(np.concatenate([ids, ids]), np.concatenate([np.full(N, FA, np.uint32), np.full(N, FS, np.uint32)]),
                  np.concatenate([ts, ts]), np.concatenate([ages, scores]), INSERT_REFERENCE, want_flags=False)


@pytest.mark.parametrize("name", ["g5_query_seeded_2k.json", "g5_query_seeded_100k.json"])
def test_scans_match_reference_queries(name):
    g = load_golden(name)
    ids, ts, ages, scores = _nodes(g["N"], g["seed"])
    ordinal = {int(v): i for i, v in enumerate(ids.tolist())}
    fld = {"age": FA, "score": FS}
    with bmx.Engine(4 * g["N"]) as e:
        _load(e, ids, ts, ages, scores)
        e.index_build(FA)                     # bullet.index('n','age')
        assert e.index_size(FA) == g["N"] and e.index_size(FS) == g["N"]
        for q in g["queries"]:
            if q["op"] == "equals":
                got = e.scan_equals(fld[q["field"]], q["args"][0])
            elif q["op"] == "range":
                got = e.scan_range(fld[q["field"]], q["args"][0], min(q["args"][1], 2**62))
            elif q["op"] == "count":
                assert e.scan_count(fld[q["field"]], q["args"][0], q["args"][0]) == q["count"]
                continue
            elif q["op"] == "filter_and":
                (a0, a1), (s0, s1) = q["args"]
                got = e.scan_filter([(FA, a0, a1), (FS, s0, s1)])
                got2 = e.scan_filter([(FS, s0, s1), (FA, a0, a1)])
                assert sorted(got.tolist()) == sorted(got2.tolist())
            ords = sorted(ordinal[int(x)] for x in got.tolist())
            assert len(ords) == q["count"], q
            assert sum(ords) == q["ordinal_sum"], q
            if "ordinals" in q:
                assert ords == sorted(q["ordinals"]), q


def test_scan_sees_fresh_state_after_merges_and_is_deterministic():
    ids, ts, ages, scores = _nodes(5000, 99)
    o = Oracle()
    with bmx.Engine(40000) as e:
        _load(e, ids, ts, ages, scores)
        o.merge_batch(np.concatenate([ids, ids]), np.concatenate([np.full(5000, FA, np.uint32), np.full(5000, FS, np.uint32)]),
                      np.concatenate([ts, ts]), np.concatenate([ages, scores]))
        a = e.scan_range(FA, 10, 20); b = e.scan_range(FA, 10, 20)
        assert np.array_equal(a, b)                                   # same order on repeat
        assert np.array_equal(np.sort(a), np.sort(o.scan_range(FA, 10, 20)))
        # update some ages with newer clocks, add new nodes; the next scan must see them (reference: a fresh index)
        upd = ids[::7]; newv = (ages[::7] + 50) % 100
        e.merge_batch(upd, np.full(len(upd), FA, np.uint32), np.full(len(upd), 1000), newv)
        o.merge_batch(upd, np.full(len(upd), FA, np.uint32), np.full(len(upd), 1000), newv)
        extra = np.array([streams.splitmix64(10**6 + i) for i in range(300)], np.uint64)
        e.merge_batch(extra, np.full(300, FA, np.uint32), np.full(300, 5), np.full(300, 15))
        o.merge_batch(extra, np.full(300, FA, np.uint32), np.full(300, 5), np.full(300, 15))
        for lo, hi in [(10, 20), (0, 99), (15, 15), (60, 61), (-5, -1), (100, 10**9), (50, 49)]:
            assert np.array_equal(np.sort(e.scan_range(FA, lo, hi)), np.sort(o.scan_range(FA, lo, hi))), (lo, hi)
            assert e.scan_count(FA, lo, hi) == o.scan_count(FA, lo, hi)
        assert e.index_size(FA) == 5300


def test_wide_values_use_the_int64_column():
    n = 3000
    ids = streams.splitmix64_np(np.arange(1, n + 1, dtype=np.uint64))
    rng = np.random.default_rng(5)
    vals = rng.integers(-(2**53 - 1), 2**53 - 1, n)
    vals[:5] = [2**31, -(2**31) - 1, 2**53 - 1, -(2**53 - 1), 0]
    o = Oracle()
    with bmx.Engine(10000) as e:
        e.merge_batch(ids, np.full(n, FS, np.uint32), np.full(n, 7), vals)
        o.merge_batch(ids, np.full(n, FS, np.uint32), np.full(n, 7), vals)
        for lo, hi in [(-(2**62), 2**62), (0, 2**53), (2**31, 2**31), (-(2**31) - 1, 2**31), (2**53 - 1, 2**53 - 1), (1, 0)]:
            assert np.array_equal(np.sort(e.scan_range(FS, lo, hi)), np.sort(o.scan_range(FS, lo, hi))), (lo, hi)


def test_empty_index_and_small_capacity():
    with bmx.Engine(1000) as e:
        assert len(e.scan_range(FA, 0, 10)) == 0 and e.scan_count(FA, 0, 10) == 0
        e.merge_batch([1, 2, 3], [FA] * 3, [5, 5, 5], [1, 2, 3])
        assert e.scan_count(FA, 2, 3) == 2
        got = e.scan_range(FA, 1, 3, cap=2)          # truncated output, full count is still reported by scan_count
        assert len(got) == 2
        with pytest.raises(bmx.BmxError):
            e.index_drop(FS + 1)
        e.index_drop(FA)
        assert e.scan_count(FA, 1, 3) == 3           # auto re-created, like the reference's equals()/range()


def test_scan_at_bench_size_properties():
    """10M-row int32 column (config 3): counts add up across a partition of the value domain; equals matches a numpy count."""
    R = 10_000_000
    ids = streams.splitmix64_np(np.arange(1, R + 1, dtype=np.uint64))
    with np.errstate(over="ignore"):
        ages = (streams.splitmix64_np(ids ^ np.uint64(0xABCDEF)) % np.uint64(100)).astype(np.int64)
    with bmx.Engine(R + 1000) as e:
        e.load_rows(ids, np.full(R, FA, np.uint32), np.full(R, 5, np.int64), ages)
        assert e.index_size(FA) == R
        parts = [(0, 9), (10, 49), (50, 98), (99, 99)]
        counts = [e.scan_count(FA, lo, hi) for lo, hi in parts]
        assert sum(counts) == R
        assert counts[3] == int((ages == 99).sum())
        got = e.scan_equals(FA, 42)
        assert len(got) == int((ages == 42).sum())
        assert set(got[:1000].tolist()) <= set(ids[ages == 42].tolist())
        assert len(np.unique(got)) == len(got)
