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


def test_scan_at_bench_size():
    """10M-row int32 column (config 3): counts add up across a partition of the value domain; equals / range id sets equal numpy's."""
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
        for lo, hi in [(42, 42), (10, 19), (0, 49)]:        # full id-set comparison with numpy: 1 %, 10 %, 50 % of the rows
            got = e.scan_range(FA, lo, hi)
            assert np.array_equal(np.sort(got), _expected_ids(ids, ages, lo, hi)), (lo, hi)


def _expected_ids(ids, vals, lo, hi):
    return np.sort(ids[(vals >= lo) & (vals <= hi)])


@pytest.mark.parametrize("wide", [False, True])
def test_large_column_scans_match_numpy(wide):
    """Index beyond 2048 scan blocks (16.78M rows): the emit pass takes EIGHT 8192-row blocks per workgroup (k_scan_emit<.., 8>) — its one-scan fast
    path (<= 8192 matches in the eight blocks) and its block-by-block fall-through, workgroups on both sides of that limit in one scan, a row count
    that is no multiple of 65536 (the last workgroup owns fewer than eight blocks, the last block is ragged). Full sorted-id-set comparison with numpy
    for ids and for positions (reference: src/bullet-query.js:186-261 on a fresh index). int32 column and the int64 column of wide values."""
    R = 20_000_003 + (4099 if wide else 0)
    assert R > 2048 * 8192 and R % 65536 != 0
    f = FS if wide else FA
    sh = 33 if wide else 0
    ids = streams.splitmix64_np(np.arange(1, R + 1, dtype=np.uint64))
    with np.errstate(over="ignore"):
        vals = ((streams.splitmix64_np(ids ^ np.uint64(0xABCDEF)) % np.uint64(1000)).astype(np.int64)) << sh
    with bmx.Engine(R + 1000) as e:
        for r0 in range(0, R, 5_000_000):
            sl = slice(r0, min(R, r0 + 5_000_000))
            e.load_rows(ids[sl], np.full(sl.stop - sl.start, f, np.uint32), np.full(sl.stop - sl.start, 5, np.int64), vals[sl])
        assert e.index_size(f) == R
        col_ids = e.index_ids(f)
        assert len(col_ids) == R and np.array_equal(np.sort(col_ids), np.sort(ids))
        # selectivity: 0.1 % and 10 % (fast path everywhere), 12.5 % (workgroups on both sides of the 8192-match limit), 20 % and 50 % (fall-through),
        # nothing, everything
        # ... and around 29.3 % (2400 matches per 8192-row block, SCAN_STREAM_MIN): from there on a block's ids are STREAMED from the id column instead of
        # gathered — 28 % (gather everywhere), 29.2 % / 29.4 % (blocks on both sides of the switch inside one workgroup), 31 % and more (streamed
        # almost everywhere; every wave's packed matches leave LDS as 16-byte stores, odd first ranks included)
        for lo, hi in [(42, 42), (100, 199), (100, 224), (300, 499), (0, 499), (2000, 3000), (-5, 5000), (100, 379), (100, 391), (100, 393), (100, 409), (100, 459), (100, 489)]:
            want = _expected_ids(ids, vals, lo << sh, hi << sh)
            got = e.scan_range(f, lo << sh, hi << sh)
            assert len(got) == len(want), (wide, lo, hi, len(got), len(want))
            assert e.scan_count(f, lo << sh, hi << sh) == len(want)
            pos = e.scan_range_pos(f, lo << sh, hi << sh)
            assert len(pos) == len(want) and (len(pos) < 2 or bool(np.all(pos[1:] > pos[:-1]))), (wide, lo, hi)
            assert np.array_equal(col_ids[pos], got), (wide, lo, hi, "positions and ids name the same rows in the same order")
            assert np.array_equal(np.sort(got), want), (wide, lo, hi)
        # truncated outputs keep the full count
        got = e.scan_range(f, 0, 499 << sh, cap=1000); pos = e.scan_range_pos(f, 0, 499 << sh, cap=1000)
        assert len(got) == 1000 and np.array_equal(col_ids[pos], got)
        # a device output buffer that starts in the UPPER half of a 16-byte unit (the streamed blocks pair their stores by address, not by rank),
        # truncated in the middle of a streamed block
        import torch
        dev = torch.device("cuda", 0)
        want = e.scan_range(f, 0, 499 << sh)
        buf = torch.zeros(len(want) + 8, dtype=torch.int64, device=dev); n_out = torch.zeros(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        for off, cap in ((1, len(want)), (0, len(want)), (1, 123_457), (2, 123_456)):
            buf.zero_(); torch.cuda.synchronize()
            e.scan_range_dev(f, 0, 499 << sh, buf[off:], cap, n_out)
            e.sync()
            assert int(n_out.item()) == len(want)
            assert np.array_equal(buf[off:off + cap].cpu().numpy().view(np.uint64), want[:cap]), (wide, off, cap)
            assert int(buf[off + cap:].abs().sum().item()) == 0 and (off == 0 or int(buf[:off].abs().sum().item()) == 0), "nothing outside [0, cap) is written"


def test_positions_on_small_and_maintained_indexes():
    """bmx_scan_range_pos / bmx_index_ids on the one-block-per-workgroup path, through the mapped-memory small-answer path and after rows were
    appended by the change log: ids[pos] always equals the id-mode answer."""
    rng = np.random.default_rng(21)
    n = 50_000
    ids = streams.splitmix64_np(np.arange(1, n + 1, dtype=np.uint64))
    vals = rng.integers(-100, 101, n).astype(np.int64)
    o = Oracle()
    with bmx.Engine(400_000) as e:
        e.load_rows(ids, np.full(n, FA, np.uint32), np.full(n, 5, np.int64), vals); o.load_rows(ids, np.full(n, FA, np.uint32), np.full(n, 5, np.int64), vals)
        e.index_build(FA)
        for b in range(3):
            m = 20_000
            rows = rng.integers(0, n + 30_000, m)
            d = (streams.splitmix64_np(rows.astype(np.uint64) + np.uint64(1)), np.full(m, FA, np.uint32), rng.integers(6, 50, m).astype(np.int64), rng.integers(-100, 101, m).astype(np.int64))
            e.merge_batch(*d); o.merge_batch(*d)
            col = e.index_ids(FA)
            assert len(col) == e.index_size(FA) == o.scan_count(FA, -2**62, 2**62)
            for lo, hi in [(-100, 100), (0, 0), (7, 30), (200, 300), (5, 4)]:
                got = e.scan_range(FA, lo, hi); pos = e.scan_range_pos(FA, lo, hi)
                assert np.array_equal(col[pos], got), (b, lo, hi)
                assert np.array_equal(np.sort(got), np.sort(o.scan_range(FA, lo, hi))), (b, lo, hi)
            assert np.array_equal(e.index_ids(FA, 10, 5), col[10:15])
        with pytest.raises(bmx.BmxError):
            e.index_ids(FA, e.index_size(FA) - 1, 5)
    o.close()
