"""GPU parity: bmx_put_rows (rows decided on the host are stored as given) and tombstones (BMX_VAL_DELETED) vs the CPU oracle.
What they restate: a node-level write replaces the node, so fields it no longer carries go away (reference src/bullet-crt.js:236-248); a `deleted`
sync entry becomes setData(path, null) (src/bullet-network-sync.js:553-555); _addToIndex skips null values (src/bullet-query.js:83-85)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import bmx
from oracle import streams
from oracle.oracle import Oracle, rows_digest, INSERT_REFERENCE, VAL_DELETED

FA, FB = streams.field_hash(1), streams.field_hash(2)


def _ids(rows):
    return streams.splitmix64_np(np.asarray(rows, dtype=np.uint64) + np.uint64(1))


def _same_state(e, o, tag):
    assert rows_digest(*e.dump_rows()) == o.digest(), tag
    assert e.row_count() == len(o), tag                      # slots in use: tombstones keep theirs
    for f in (FA, FB):
        for lo, hi in [(-2**62, 2**62), (-50, 50), (0, 0), (-(2**31), -(2**31)), (7, 3)]:
            assert np.array_equal(np.sort(e.scan_range(f, lo, hi)), np.sort(o.scan_range(f, lo, hi))), (tag, f, lo, hi)
            assert e.scan_count(f, lo, hi) == o.scan_count(f, lo, hi), (tag, f, lo, hi)
    terms = [(FA, -100, 100), (FB, -100, 100)]
    assert np.array_equal(np.sort(e.scan_filter(terms)), np.sort(o.scan_filter_and(terms))), tag


@pytest.mark.parametrize("n,indexed", [(40, True), (5000, True), (200_000, True), (5000, False)])
def test_put_and_tombstones_match_the_oracle(n, indexed):
    rng = np.random.default_rng(n + indexed)
    ids = _ids(np.arange(n))
    e = bmx.Engine(capacity_rows=6 * n + 100_000, flags=bmx.CTX_FIXED_CAPACITY); o = Oracle()
    for f in (FA, FB):
        v = rng.integers(-100, 101, n).astype(np.int64)
        e.load_rows(ids, np.full(n, f, np.uint32), np.full(n, 50, np.int64), v); o.load_rows(ids, np.full(n, f, np.uint32), np.full(n, 50, np.int64), v)
    if indexed:
        e.index_build(FA); e.index_build(FB)
    _same_state(e, o, "loaded")
    full0, _ = e.index_refresh_counts()
    # 1. puts: lower clocks than the resident ones (a merge would reject them), unique keys: overwritten as given
    k = max(4, n // 3)
    sel = rng.choice(n, k, replace=False)
    d = (ids[sel], np.where(rng.random(k) < 0.5, FA, FB).astype(np.uint32), rng.integers(1, 40, k).astype(np.int64), rng.integers(-100, 101, k).astype(np.int64))
    e.put_rows(*d); o.put_rows(*d)
    _same_state(e, o, "put")
    # 2. tombstones: existing rows, and keys that never existed (the key takes a slot, no scan ever sees it)
    sel = rng.choice(n, k, replace=False)
    dead_ids = np.concatenate([ids[sel], _ids(np.arange(10 * n, 10 * n + k // 2))])
    m = len(dead_ids)
    d = (dead_ids, np.full(m, FA, np.uint32), rng.integers(45, 60, m).astype(np.int64), np.full(m, VAL_DELETED, np.int64))
    e.put_rows(*d); o.put_rows(*d)
    _same_state(e, o, "tombstones")
    ts, val, found = e.get_rows(dead_ids[:3], np.full(3, FA, np.uint32))
    assert found.all() and (val == VAL_DELETED).all() and np.array_equal(ts, d[2][:3])
    # 3. merges against tombstones: a delta at the tombstone's ts or later brings the row back, an older one is historical; duplicates included
    back = rng.choice(m, m // 2, replace=False)
    bid = np.concatenate([dead_ids[back], dead_ids[back][: m // 8]])
    bts = np.concatenate([d[2][back] + rng.integers(-3, 4, len(back)), d[2][back][: m // 8] + 5])
    bval = rng.integers(-100, 101, len(bid)).astype(np.int64)
    applied, _, st = e.merge_batch(bid, np.full(len(bid), FA, np.uint32), bts, bval)
    _, want = o.merge_batch(bid, np.full(len(bid), FA, np.uint32), bts, bval, INSERT_REFERENCE)
    assert np.array_equal(applied, want)
    _same_state(e, o, "merge over tombstones")
    # 4. a real -2^31 next to tombstones: the int32 column cannot tell them apart, the index goes wide and stays right
    d = (ids[:2], np.full(2, FA, np.uint32), np.full(2, 99, np.int64), np.array([-(2**31), 2**31], np.int64))
    e.put_rows(*d); o.put_rows(*d)
    _same_state(e, o, "int32 limits")
    # 5. a node replaced by an object with fewer fields (FB gone), then deleted altogether, then written again
    node = ids[5:6]
    for step, rows in enumerate([[(FA, 7, 11), (FB, 7, VAL_DELETED)], [(FA, 8, VAL_DELETED), (FB, 8, VAL_DELETED)], [(FA, 9, 13), (FB, 9, 14)]]):
        d = (np.repeat(node, len(rows)), np.array([r[0] for r in rows], np.uint32), np.array([r[1] for r in rows], np.int64), np.array([r[2] for r in rows], np.int64))
        e.put_rows(*d); o.put_rows(*d)
        _same_state(e, o, ("node", step))
    if indexed:
        full1, inc1 = e.index_refresh_counts()
        assert full1 - full0 <= 2, "puts and tombstones are applied to the maintained indexes from the change log (one rebuild per index at most: the int32 -> int64 switch)"
    with pytest.raises(bmx.BmxError):      # the tombstone value is not a value a merge may carry
        e.merge_batch(ids[:1], [FA], [100], [VAL_DELETED])
    e.close(); o.close()
