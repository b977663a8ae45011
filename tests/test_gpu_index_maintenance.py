"""GPU parity: incremental index maintenance (the device-side _updateIndices, reference src/bullet-query.js:82-110). After every merge the
scans over maintained indexes equal the oracle's scans of the same table, and the engine reports that it applied its change log instead of
rebuilding — except where a rebuild is the contract (table growth, strict-flag / unique-key merges), after which maintenance resumes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import bmx
from oracle import streams
from oracle.oracle import Oracle, INSERT_REFERENCE, INSERT_DELTA

FA, FB, FC = streams.field_hash(1), streams.field_hash(2), streams.field_hash(3)


def _ids(rows):
    return streams.splitmix64_np(np.asarray(rows, dtype=np.uint64) + np.uint64(1))


def _check_scans(e, o, rng, span, tag):
    for f in (FA, FB):
        for _ in range(4):
            lo = int(rng.integers(-span - 2, span + 3)); hi = lo + int(rng.integers(-1, span + 3))
            got = np.sort(e.scan_range(f, lo, hi)); ref = np.sort(o.scan_range(f, lo, hi))
            assert np.array_equal(got, ref), (tag, f, lo, hi, len(got), len(ref))
            assert e.scan_count(f, lo, hi) == len(ref), (tag, f, lo, hi)
    assert e.index_size(FA) == o.scan_count(FA, -2**62, 2**62), tag
    terms = [(FA, -span // 2, span // 2), (FB, 0, span)]
    assert np.array_equal(np.sort(e.scan_filter(terms)), np.sort(o.scan_filter_and(terms))), tag


def _batch(rng, n, nodes, new_from, span, tmax):
    """updates of existing nodes (three fields, duplicates and ties are common) + rows of new nodes"""
    rows = rng.integers(0, nodes, n)
    new = rng.random(n) < 0.15
    rows[new] = new_from + rng.integers(0, max(n // 4, 1), int(new.sum()))
    ids = _ids(rows)
    fields = np.array([FA, FB, FC], np.uint32)[rng.integers(0, 3, n)]
    ts = rng.integers(1, tmax, n).astype(np.int64)
    val = rng.integers(-span, span + 1, n).astype(np.int64)
    return ids, fields, ts, val


@pytest.mark.parametrize("seed,nodes,nb", [(0, 3000, 10), (1, 60000, 8), (2, 200000, 6)])
def test_scans_after_every_merge_use_the_change_log(seed, nodes, nb):
    rng = np.random.default_rng(700 + seed)
    span = 1000
    e = bmx.Engine(capacity_rows=8 * nodes + 2_000_000, flags=bmx.CTX_FIXED_CAPACITY); o = Oracle()
    ids = _ids(np.arange(nodes))
    for f in (FA, FB):
        v = rng.integers(-span, span + 1, nodes).astype(np.int64)
        e.load_rows(ids, np.full(nodes, f, np.uint32), np.full(nodes, 5, np.int64), v); o.load_rows(ids, np.full(nodes, f, np.uint32), np.full(nodes, 5, np.int64), v)
    e.index_build(FA); e.index_build(FB)
    _check_scans(e, o, rng, span, "fresh")
    full0, inc0 = e.index_refresh_counts()
    new_from = nodes
    for b in range(nb):
        for sub in range(int(rng.integers(1, 4))):          # one to three merges between two scans
            n = int(rng.choice([1, 300, 5000, 40000]))
            mode = [INSERT_REFERENCE, INSERT_DELTA][int(rng.integers(0, 2))]
            d = _batch(rng, n, nodes, new_from, span, 10 + 3 * b)
            new_from += n
            applied, _, _ = e.merge_batch(*d, insert_mode=mode)
            _, ow = o.merge_batch(*d, mode)
            assert np.array_equal(applied, ow), (seed, b, sub)
        _check_scans(e, o, rng, span, (seed, b))
    full1, inc1 = e.index_refresh_counts()
    assert full1 == full0, "a maintained index was rebuilt from the table (%d -> %d full builds)" % (full0, full1)
    assert inc1 - inc0 == nb, (inc0, inc1)
    e.close(); o.close()


def test_values_leaving_int32_and_records_path():
    rng = np.random.default_rng(11)
    nodes = 20000
    e = bmx.Engine(capacity_rows=8 * nodes + 2_000_000, flags=bmx.CTX_FIXED_CAPACITY); o = Oracle()
    ids = _ids(np.arange(nodes))
    v = rng.integers(-100, 101, nodes).astype(np.int64)
    for f in (FA, FB):
        e.load_rows(ids, np.full(nodes, f, np.uint32), np.full(nodes, 5, np.int64), v); o.load_rows(ids, np.full(nodes, f, np.uint32), np.full(nodes, 5, np.int64), v)
    e.index_build(FA); e.index_build(FB)
    full0, _ = e.index_refresh_counts()
    # a batch in the exchange's record format (device pointers), with values beyond 32 bits: the int32 column stops being usable
    n = 4096
    d = _batch(rng, n, nodes, nodes, 100, 50)
    d[3][::7] = (2**40 + rng.integers(0, 1000, len(d[3][::7]))).astype(np.int64)
    recs = np.zeros((n, 4), np.int64)
    recs[:, 0] = d[0].view(np.int64); recs[:, 1] = d[1].astype(np.int64); recs[:, 2] = d[2]; recs[:, 3] = d[3]
    dev = torch.device("cuda", 0)
    rt = torch.from_numpy(recs).to(dev); applied = torch.zeros(n, dtype=torch.int32, device=dev); na = torch.zeros(1, dtype=torch.int64, device=dev)
    e.merge_records_dev(n, rt, INSERT_REFERENCE, applied=applied, n_applied=na); e.sync()
    _, ow = o.merge_batch(*d, INSERT_REFERENCE)
    assert np.array_equal(applied[:int(na.item())].cpu().numpy().astype(np.uint32), ow)
    _check_scans(e, o, rng, 100, "wide")
    got = np.sort(e.scan_range(FA, 2**40, 2**41)); ref = np.sort(o.scan_range(FA, 2**40, 2**41))
    assert len(ref) > 0 and np.array_equal(got, ref)
    assert e.index_refresh_counts()[0] == full0          # leaving int32 only switches the scans to the int64 column
    d = _batch(rng, n, nodes, nodes + n, 100, 60)
    full1 = e.index_refresh_counts()[0]
    e.merge_batch(*d); o.merge_batch(*d, INSERT_REFERENCE)
    _check_scans(e, o, rng, 100, "wide, next batch")
    assert e.index_refresh_counts()[0] == full1
    e.close(); o.close()


def test_rebuild_where_it_is_the_contract_then_maintenance_resumes():
    rng = np.random.default_rng(23)
    nodes = 5000
    e = bmx.Engine(capacity_rows=2 * nodes); o = Oracle()           # small and growable: the table is rehashed along the way
    ids = _ids(np.arange(nodes))
    v = rng.integers(-500, 501, nodes).astype(np.int64)
    for f in (FA, FB):
        e.load_rows(ids, np.full(nodes, f, np.uint32), np.full(nodes, 5, np.int64), v); o.load_rows(ids, np.full(nodes, f, np.uint32), np.full(nodes, 5, np.int64), v)
    e.index_build(FA); e.index_build(FB)
    new_from = nodes
    for b in range(14):
        n = 3000
        d = _batch(rng, n, nodes, new_from, 500, 20 + b); new_from += n
        kind = ["plain", "strict", "plain", "unique-free plain", "plain"][b % 5]
        full_a, inc_a = e.index_refresh_counts()
        if b == 7:
            e.reserve(200_000)                      # rehash into a larger table: every row moves
        if kind == "strict":
            applied, _, _ = e.merge_batch(*d, insert_mode=INSERT_REFERENCE | bmx.MERGE_STRICT_FLAGS, want_flags=True)
        else:
            applied, _, _ = e.merge_batch(*d, insert_mode=INSERT_REFERENCE)
        _, ow = o.merge_batch(*d, INSERT_REFERENCE)
        assert np.array_equal(applied, ow), b
        _check_scans(e, o, rng, 500, ("mixed", b))
        full_b, inc_b = e.index_refresh_counts()
        if kind == "strict":
            assert full_b > full_a, "a strict-flags merge does not log: the indexes have to be rebuilt"
        if b == 7:
            assert full_b > full_a, "after a rehash the recorded slot positions are void: the indexes have to be rebuilt"
        if b in (9, 10, 13):
            assert full_b == full_a and inc_b > inc_a, "maintenance did not resume after the rebuild (batch %d)" % b
    assert e.index_refresh_counts()[1] > 0, "maintenance never resumed"
    e.close(); o.close()


def test_index_created_while_another_is_being_maintained():
    rng = np.random.default_rng(5)
    nodes = 10000
    e = bmx.Engine(capacity_rows=8 * nodes + 2_000_000, flags=bmx.CTX_FIXED_CAPACITY); o = Oracle()
    ids = _ids(np.arange(nodes))
    v = rng.integers(-50, 51, nodes).astype(np.int64)
    for f in (FA, FB):
        e.load_rows(ids, np.full(nodes, f, np.uint32), np.full(nodes, 5, np.int64), v); o.load_rows(ids, np.full(nodes, f, np.uint32), np.full(nodes, 5, np.int64), v)
    e.index_build(FA)
    d = _batch(rng, 5000, nodes, nodes, 50, 30)
    e.merge_batch(*d); o.merge_batch(*d, INSERT_REFERENCE)
    # FB is indexed only now (auto-created by its first scan), with FA's log entries pending
    got = np.sort(e.scan_range(FB, -10, 10)); ref = np.sort(o.scan_range(FB, -10, 10))
    assert np.array_equal(got, ref)
    _check_scans(e, o, rng, 50, "late index")
    d = _batch(rng, 5000, nodes, nodes + 5000, 50, 40)
    e.merge_batch(*d); o.merge_batch(*d, INSERT_REFERENCE)
    _check_scans(e, o, rng, 50, "late index, next batch")
    # dropping and re-creating an index while the log is live
    e.index_drop(FA)
    d = _batch(rng, 5000, nodes, nodes + 10000, 50, 50)
    e.merge_batch(*d); o.merge_batch(*d, INSERT_REFERENCE)
    _check_scans(e, o, rng, 50, "after drop")
    e.close(); o.close()


N_FUZZ = int(__import__("os").environ.get("BMX_FUZZ_SEEDS", "6"))


@pytest.mark.parametrize("seed", range(N_FUZZ))
def test_random_sequences_of_merges_index_changes_and_scans(seed):
    """Whatever mixes of merges (all modes), index builds/drops, table growth and scans: every scan equals the oracle's."""
    rng = np.random.default_rng(9000 + seed)
    nodes = int(rng.choice([200, 5000, 40000]))
    span = int(rng.choice([5, 1000]))
    fixed = bool(rng.integers(0, 2))
    e = bmx.Engine(capacity_rows=(8 * nodes + 600_000) if fixed else max(64, nodes // 2), flags=bmx.CTX_FIXED_CAPACITY if fixed else 0); o = Oracle()
    ids = _ids(np.arange(nodes))
    for f in (FA, FB):
        v = rng.integers(-span, span + 1, nodes).astype(np.int64)
        e.load_rows(ids, np.full(nodes, f, np.uint32), np.full(nodes, 5, np.int64), v); o.load_rows(ids, np.full(nodes, f, np.uint32), np.full(nodes, 5, np.int64), v)
    new_from = nodes
    have = set()
    for step in range(40):
        op = rng.choice(["merge", "merge", "merge", "delta", "strict", "unique", "scan", "scan", "build", "drop", "reserve"])
        if op in ("merge", "delta", "strict"):
            n = int(rng.choice([1, 64, 700, 9000]))
            d = _batch(rng, n, nodes, new_from, span, 10 + step); new_from += n
            mode = INSERT_DELTA if op == "delta" else INSERT_REFERENCE
            flags = bmx.MERGE_STRICT_FLAGS if op == "strict" else 0
            applied, _, _ = e.merge_batch(*d, insert_mode=mode | flags, want_flags=bool(flags))
            _, ow = o.merge_batch(*d, mode)
            assert np.array_equal(applied, ow), (seed, step, op)
        elif op == "unique":
            n = int(rng.choice([1, 300, 5000]))
            rows = rng.permutation(nodes)[:min(n, nodes)]
            d = (_ids(rows), np.full(len(rows), FA, np.uint32), rng.integers(1, 60, len(rows)).astype(np.int64), rng.integers(-span, span + 1, len(rows)).astype(np.int64))
            applied, _, _ = e.merge_batch(*d, insert_mode=INSERT_REFERENCE | bmx.MERGE_UNIQUE_KEYS)
            _, ow = o.merge_batch(*d, INSERT_REFERENCE)
            assert np.array_equal(applied, ow), (seed, step, op)
        elif op == "build":
            f = [FA, FB][int(rng.integers(0, 2))]; e.index_build(f); have.add(f)
        elif op == "drop" and have:
            f = sorted(have)[int(rng.integers(0, len(have)))]; e.index_drop(f); have.discard(f)
        elif op == "reserve" and not fixed:
            e.reserve(int(e.row_count() * 1.5) + 1000)
        else:
            _check_scans(e, o, rng, span, (seed, step)); have.update((FA, FB))
    _check_scans(e, o, rng, span, (seed, "end"))
    e.close(); o.close()


def test_more_indexes_than_the_maintenance_pass_handles_and_appends_beyond_the_head_room():
    """Nine indexes: beyond the eight the change-log pass keeps up to date, every stale index is rebuilt instead — scans stay right. And a batch that
    creates more rows than an index column has head room for (n/8 + 64K) makes that index rebuild once, after which it is maintained again."""
    rng = np.random.default_rng(3)
    fields = [streams.field_hash(10 + k) for k in range(9)]
    nodes = 3000
    e = bmx.Engine(capacity_rows=2_000_000, flags=bmx.CTX_FIXED_CAPACITY); o = Oracle()
    ids = _ids(np.arange(nodes))
    for f in fields:
        v = rng.integers(-50, 51, nodes).astype(np.int64)
        e.load_rows(ids, np.full(nodes, f, np.uint32), np.full(nodes, 5, np.int64), v); o.load_rows(ids, np.full(nodes, f, np.uint32), np.full(nodes, 5, np.int64), v)
    for f in fields[:8]:
        e.index_build(f)

    def check(tag, fs):
        for f in fs:
            got = np.sort(e.scan_range(f, -20, 20)); ref = np.sort(o.scan_range(f, -20, 20))
            assert np.array_equal(got, ref), (tag, f)

    for b in range(3):
        n = 4000
        rows = rng.integers(0, nodes + 500, n)
        d = (_ids(rows), np.array(fields, np.uint32)[rng.integers(0, 9, n)], rng.integers(6, 40, n).astype(np.int64), rng.integers(-50, 51, n).astype(np.int64))
        e.merge_batch(*d); o.merge_batch(*d, INSERT_REFERENCE)
        check(("eight", b), fields[:8])
    assert e.index_refresh_counts()[1] >= 3                   # eight indexes: maintained
    e.index_build(fields[8])                                   # the ninth
    for b in range(3):
        n = 4000
        rows = rng.integers(0, nodes + 2000, n)
        d = (_ids(rows), np.array(fields, np.uint32)[rng.integers(0, 9, n)], rng.integers(40, 80, n).astype(np.int64), rng.integers(-50, 51, n).astype(np.int64))
        e.merge_batch(*d); o.merge_batch(*d, INSERT_REFERENCE)
        check(("nine", b), fields)
    for f in fields[1:]:
        e.index_drop(f)
    check("one left", fields[:1])
    # far more new rows of the indexed field than its columns have room for: 3000 rows + 64K + n/8 head room < 120000 new rows
    full0, inc0 = e.index_refresh_counts()
    n = 120_000
    d = (_ids(np.arange(10_000_000, 10_000_000 + n)), np.full(n, fields[0], np.uint32), np.full(n, 7, np.int64), rng.integers(-50, 51, n).astype(np.int64))
    e.merge_batch(*d); o.merge_batch(*d, INSERT_REFERENCE)
    check("overflow", fields[:1])
    full1, _ = e.index_refresh_counts()
    assert full1 > full0, "the appended rows cannot have fitted: the index must have been rebuilt"
    d = (_ids(np.arange(10_000_000, 10_000_000 + 500)), np.full(500, fields[0], np.uint32), np.full(500, 9, np.int64), rng.integers(-50, 51, 500).astype(np.int64))
    e.merge_batch(*d); o.merge_batch(*d, INSERT_REFERENCE)
    check("after overflow", fields[:1])
    assert e.index_refresh_counts()[0] == full1                # maintained again
    e.close(); o.close()


def test_overflow_of_an_index_that_was_not_the_one_asked_for():
    """Two maintained indexes share the change log. A merge creates more rows of B's field than B's columns have head room for; then A is scanned
    (which applies the log to BOTH and forgets it), then B. B must come back through a rebuild — it can never be refreshed from a later log, whose
    entries no longer hold the rows it missed (bmx.hip refresh_from_log)."""
    rng = np.random.default_rng(11)
    nodes = 3000
    e = bmx.Engine(capacity_rows=2_000_000, flags=bmx.CTX_FIXED_CAPACITY); o = Oracle()
    ids = _ids(np.arange(nodes))
    for f in (FA, FB):
        v = rng.integers(-50, 51, nodes).astype(np.int64)
        e.load_rows(ids, np.full(nodes, f, np.uint32), np.full(nodes, 5, np.int64), v); o.load_rows(ids, np.full(nodes, f, np.uint32), np.full(nodes, 5, np.int64), v)
    e.index_build(FA); e.index_build(FB)

    def check(tag, f):
        got = np.sort(e.scan_range(f, -20, 20)); ref = np.sort(o.scan_range(f, -20, 20))
        assert np.array_equal(got, ref), (tag, f, len(got), len(ref))
        assert e.index_size(f) == o.scan_count(f, -2**62, 2**62), (tag, f)

    for variant in ("scan B right away", "another merge before B is scanned"):
        n = 120_000                                           # B: 3000 (+ earlier) rows + 64K + n/8 head room < 120000 new rows
        base = 10_000_000 if variant.startswith("scan") else 20_000_000
        d = (_ids(np.arange(base, base + n)), np.full(n, FB, np.uint32), np.full(n, 7, np.int64), rng.integers(-50, 51, n).astype(np.int64))
        e.merge_batch(*d); o.merge_batch(*d, INSERT_REFERENCE)
        full0, _ = e.index_refresh_counts()
        check((variant, "A"), FA)                             # brings A up to date from the log; B's appended rows do not fit
        if not variant.startswith("scan"):
            m = 700                                           # new log entries behind the ones B missed
            d = (_ids(np.concatenate([np.arange(base, base + m // 2), np.arange(30_000_000, 30_000_000 + m // 2)])), np.full(m, FB, np.uint32),
                 np.full(m, 9, np.int64), rng.integers(-50, 51, m).astype(np.int64))
            e.merge_batch(*d); o.merge_batch(*d, INSERT_REFERENCE)
        check((variant, "B"), FB)
        assert e.index_refresh_counts()[0] > full0, "B cannot have been maintained through the overflow: it must have been rebuilt"
        check((variant, "A again"), FA)
    # and with nothing left to maintain the merges stop logging; a new index starts from a fresh build
    e.index_drop(FA); e.index_drop(FB)
    d = (_ids(np.arange(40_000_000, 40_000_500)), np.full(500, FA, np.uint32), np.full(500, 7, np.int64), rng.integers(-50, 51, 500).astype(np.int64))
    e.merge_batch(*d); o.merge_batch(*d, INSERT_REFERENCE)
    check("after dropping every index", FA)
    e.close(); o.close()
