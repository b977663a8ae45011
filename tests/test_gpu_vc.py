"""GPU parity, N4: K-writer vector-clock rows on the device (bmx_vc_* through the C ABI) vs
 - the golden vectors made by the real reference (tests/golden/g6_vc_*.json, oracle/gen_golden.js runVcStream), and
 - oracle/bmx_oracle.c orc_vc_* on seeded random batches (hot keys, inserts, many batches, epoch wrap).
Bit-exact: flags per delta, the ascending list of last-updating deltas, and every row's (clock, value, sparse/dense)."""
import base64

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import bmx
from oracle import streams
from oracle.oracle import OracleVC
from helpers import load_golden

F0 = streams.field_hash(0)
keyid = lambda row: streams.splitmix64(row + 1)


def _compare_rows(e, o, ids, fields):
    clocks, val, st = e.get_rows(ids, fields)
    for k, (i, f) in enumerate(zip(ids, fields)):
        ref = o.get_row(int(i), int(f))
        if ref is None:
            assert st[k] == bmx.VC_ABSENT, (k, st[k])
            continue
        c, v, sparse = ref
        assert st[k] == (bmx.VC_SPARSE if sparse else bmx.VC_DENSE), (k, st[k], sparse)
        assert clocks[k].tolist() == c and val[k] == v, (k, clocks[k], c, val[k], v)


@pytest.mark.parametrize("name", ["g6_vc_unique_2k.json", "g6_vc_dups_500.json", "g6_vc_empty_start.json"])
def test_vc_device_matches_reference_golden(name):
    g = load_golden(name)
    K = len(g["writers"]); local = g["writers"].index("w")
    e = bmx.EngineVC(max(4096, 2 * (len(g["resident"]) + len(g["deltas"]))), K, local)
    if g["resident"]:
        e.load_rows([keyid(r[0]) for r in g["resident"]], [F0] * len(g["resident"]), [r[1] for r in g["resident"]], [r[2] for r in g["resident"]])
    d = g["deltas"]
    flags, upd = e.merge_batch([keyid(x[0]) for x in d], [F0] * len(d), [x[1] for x in d], [x[2] for x in d])
    assert np.array_equal(flags, np.frombuffer(base64.b64decode(g["flags_b64"]), dtype=np.uint8))
    assert upd.tolist() == g["updated"]
    assert e.row_count() == len(g["final_rows"])
    ids = np.array([keyid(r[0]) for r in g["final_rows"]], np.uint64)
    clocks, val, st = e.get_rows(ids, np.full(len(ids), F0, np.uint32))
    for k, (row, clock, v, nkeys) in enumerate(g["final_rows"]):
        assert clocks[k].tolist() == clock and val[k] == v, (row, clocks[k], clock, val[k], v)
        assert st[k] == (bmx.VC_SPARSE if nkeys == 1 else bmx.VC_DENSE), (row, st[k], nkeys)
    e.close()


@pytest.mark.parametrize("name", ["g11_vc_keysets_2k.json", "g11_vc_keysets_hot.json"])
def test_vc_device_matches_reference_on_arbitrary_key_sets(name):
    """Clocks over ordered subsets of the writers, made by the real reference: flags, updating deltas, and every final row's counters, value and
    key order (bmx_vc_*_ks)."""
    g = load_golden(name)
    K = len(g["writers"]); local = g["writers"].index("w")
    e = bmx.EngineVC(max(4096, 2 * (len(g["resident"]) + len(g["deltas"]))), K, local)
    r = g["resident"]
    if r:
        e.load_rows([keyid(x[0]) for x in r], [F0] * len(r), [x[2] for x in r], [x[3] for x in r], keysets=[bmx.keyset(x[1]) for x in r])
    d = g["deltas"]
    flags, upd = e.merge_batch([keyid(x[0]) for x in d], [F0] * len(d), [x[2] for x in d], [x[3] for x in d], keysets=[bmx.keyset(x[1]) for x in d])
    assert np.array_equal(flags, np.frombuffer(base64.b64decode(g["flags_b64"]), dtype=np.uint8))
    assert upd.tolist() == g["updated"]
    assert e.row_count() == len(g["final_rows"])
    ids = np.array([keyid(x[0]) for x in g["final_rows"]], np.uint64)
    clocks, val, st, ks = e.get_rows(ids, np.full(len(ids), F0, np.uint32), with_keysets=True)
    for k, (row, keys, clock, v) in enumerate(g["final_rows"]):
        assert clocks[k].tolist() == clock and val[k] == v and bmx.keyset_writers(ks[k]) == keys, (row, clocks[k], clock, val[k], v, hex(ks[k]), keys)
    e.close()


def _rand_keysets(rng, n, K, clocks, full=0.3):
    """random ordered subsets of the K writers; counters of the writers a clock does not name are zeroed in place"""
    ks = np.zeros(n, np.uint32)
    for j in range(n):
        order = rng.permutation(K)
        cnt = K if rng.random() < full else int(rng.integers(0, K + 1))
        keys = order[:cnt].tolist()
        mask = np.zeros(K, bool); mask[keys] = True
        clocks[j, ~mask] = 0
        ks[j] = bmx.keyset(keys)
    return ks


@pytest.mark.parametrize("K,local", [(1, 0), (2, 0), (3, 2), (8, 5)])
def test_vc_random_key_sets_match_oracle(K, local):
    """seeded batches with hot keys (lists longer than the in-lane limit go through k_vc_resolve_long), a keyed preload, several batches"""
    rng = np.random.default_rng(300 + K)
    e = bmx.EngineVC(40000, K, local); o = OracleVC(K, local)
    ids, fields, clocks, val = _rand_batch(rng, 3000, 2000, K, 3, 3)
    ks = _rand_keysets(rng, len(ids), K, clocks)
    e.load_rows(ids, fields, clocks, val, keysets=ks); o.load_rows(ids, fields, clocks, val, keysets=ks)
    seen = set(zip(ids.tolist(), fields.tolist()))
    for b in range(5):
        ids, fields, clocks, val = _rand_batch(rng, 5000, 3000, K, 3 + 5 * b, 3, hot=0.3 if b % 2 else 0.0)
        ks = _rand_keysets(rng, len(ids), K, clocks)
        f1, u1 = e.merge_batch(ids, fields, clocks, val, keysets=ks)
        f2, u2 = o.merge_batch(ids, fields, clocks, val, keysets=ks)
        assert np.array_equal(f1, f2), (b, np.nonzero(f1 != f2)[0][:10])
        assert np.array_equal(u1, u2), b
        assert e.row_count() == len(o)
        seen.update(zip(ids.tolist(), fields.tolist()))
    keys = sorted(seen)
    kid = np.array([k[0] for k in keys], np.uint64); kf = np.array([k[1] for k in keys], np.uint32)
    clocks, val, st, ks = e.get_rows(kid, kf, with_keysets=True)
    for k in range(len(keys)):
        c, v, sparse, oks = o.get_row(int(kid[k]), int(kf[k]), with_keyset=True)
        assert clocks[k].tolist() == c and val[k] == v and int(ks[k]) == oks, (k, clocks[k], c, val[k], v, hex(int(ks[k])), hex(oks))
    e.close()


def test_vc_malformed_key_sets_are_refused():
    K, local = 3, 0
    ids = np.array([keyid(1)], np.uint64); f = np.array([F0], np.uint32); val = np.array([1], np.int64)
    for keys, clock in [([3], [0, 0, 0]), ([1, 1], [0, 1, 0]), ([0], [1, 1, 0])]:      # unknown writer, a writer twice, a counter of an unnamed writer
        e = bmx.EngineVC(4096, K, local)
        with pytest.raises(bmx.BmxError) as ei:
            e.merge_batch(ids, f, np.array([clock], np.uint32), val, keysets=[bmx.keyset(keys)])
        assert ei.value.code == bmx.ERR_RANGE
        e.close()
    e = bmx.EngineVC(4096, K, local)
    with pytest.raises(bmx.BmxError):
        e.merge_batch(ids, f, np.array([[0, 0, 0]], np.uint32), val, keysets=[0xFFFFF0F1])        # a key behind the end mark
    e.close()


def _rand_batch(rng, n, nkeys, K, cmax, vr, hot=0.0, nfields=2):
    rows = rng.integers(0, nkeys, n)
    if hot > 0:
        h = rng.random(n) < hot
        rows[h] = rng.integers(0, 4, int(h.sum()))
    ids = np.array([keyid(int(r)) for r in rows], np.uint64)
    fields = np.array([streams.field_hash(int(x)) for x in rng.integers(0, nfields, n)], np.uint32)
    clocks = rng.integers(0, cmax + 1, (n, K)).astype(np.uint32)
    val = rng.integers(-vr, vr + 1, n).astype(np.int64)
    return ids, fields, clocks, val


@pytest.mark.parametrize("K,local", [(1, 0), (2, 1), (3, 2), (8, 5)])
def test_vc_random_batches_match_oracle(K, local):
    rng = np.random.default_rng(100 + K)
    e = bmx.EngineVC(40000, K, local); o = OracleVC(K, local)
    seen = set()
    for b in range(6):
        ids, fields, clocks, val = _rand_batch(rng, 5000, 3000, K, 3 + b, 3, hot=0.2 if b % 2 else 0.0)
        f1, u1 = e.merge_batch(ids, fields, clocks, val)
        f2, u2 = o.merge_batch(ids, fields, clocks, val)
        assert np.array_equal(f1, f2), (b, np.nonzero(f1 != f2)[0][:10])
        assert np.array_equal(u1, u2), b
        assert e.row_count() == len(o)
        seen.update(zip(ids.tolist(), fields.tolist()))
    keys = sorted(seen)
    _compare_rows(e, o, np.array([k[0] for k in keys], np.uint64), np.array([k[1] for k in keys], np.uint32))
    e.close()


def test_vc_all_deltas_one_key_and_absent_lookup():
    K, local = 3, 0
    rng = np.random.default_rng(7)
    n = 3000
    e = bmx.EngineVC(8192, K, local); o = OracleVC(K, local)
    ids = np.full(n, keyid(42), np.uint64); fields = np.full(n, F0, np.uint32)
    clocks = rng.integers(0, 50, (n, K)).astype(np.uint32); val = rng.integers(-5, 6, n).astype(np.int64)
    f1, u1 = e.merge_batch(ids, fields, clocks, val)
    f2, u2 = o.merge_batch(ids, fields, clocks, val)
    assert np.array_equal(f1, f2) and np.array_equal(u1, u2) and len(u1) == 1
    _compare_rows(e, o, np.array([keyid(42), keyid(43)], np.uint64), np.array([F0, F0], np.uint32))
    e.close()


def test_vc_load_rows_last_wins_then_merge():
    K, local = 3, 2
    rng = np.random.default_rng(11)
    e = bmx.EngineVC(20000, K, local); o = OracleVC(K, local)
    ids, fields, clocks, val = _rand_batch(rng, 6000, 2000, K, 5, 4)      # duplicates inside the preload: the last one stays
    e.load_rows(ids, fields, clocks, val); o.load_rows(ids, fields, clocks, val)
    assert e.row_count() == len(o)
    ids2, fields2, clocks2, val2 = _rand_batch(rng, 6000, 2500, K, 6, 4)
    f1, u1 = e.merge_batch(ids2, fields2, clocks2, val2); f2, u2 = o.merge_batch(ids2, fields2, clocks2, val2)
    assert np.array_equal(f1, f2) and np.array_equal(u1, u2)
    keys = sorted(set(zip(ids.tolist(), fields.tolist())) | set(zip(ids2.tolist(), fields2.tolist())))
    _compare_rows(e, o, np.array([k[0] for k in keys], np.uint64), np.array([k[1] for k in keys], np.uint32))
    e.close()


def test_vc_epoch_wrap_many_small_batches():
    K, local = 2, 0
    rng = np.random.default_rng(5)
    e = bmx.EngineVC(4096, K, local); o = OracleVC(K, local)
    for b in range(300):          # > 255 batches: claim tags wrap once
        ids, fields, clocks, val = _rand_batch(rng, 64, 200, K, 2 + b // 20, 2, hot=0.3, nfields=1)
        f1, u1 = e.merge_batch(ids, fields, clocks, val); f2, u2 = o.merge_batch(ids, fields, clocks, val)
        assert np.array_equal(f1, f2) and np.array_equal(u1, u2), b
    ids = np.array([keyid(r) for r in range(200)], np.uint64)
    _compare_rows(e, o, ids, np.full(200, F0, np.uint32))
    e.close()


def test_vc_errors():
    with pytest.raises(bmx.BmxError):
        bmx.EngineVC(100, 9, 0)
    with pytest.raises(bmx.BmxError):
        bmx.EngineVC(100, 3, 3)
    e = bmx.EngineVC(2048, 2, 0)
    f, u = e.merge_batch([], [], np.zeros((0, 2), np.uint32), [])
    assert len(f) == 0 and len(u) == 0
    with pytest.raises(bmx.BmxError) as ei:
        e.merge_batch([keyid(1)], [F0], [[1, 1]], [1 << 60])
    assert ei.value.code == bmx.ERR_RANGE
    e.close()


def test_vc_table_grows():
    K, local = 3, 1
    rng = np.random.default_rng(21)
    e = bmx.EngineVC(1000, K, local); o = OracleVC(K, local)      # 4096 slots to start with
    seen = set()
    for b in range(5):
        ids, fields, clocks, val = _rand_batch(rng, 20000, 60000, K, 4, 3)
        f1, u1 = e.merge_batch(ids, fields, clocks, val); f2, u2 = o.merge_batch(ids, fields, clocks, val)
        assert np.array_equal(f1, f2) and np.array_equal(u1, u2), b
        assert e.row_count() == len(o)
        seen.update(zip(ids.tolist(), fields.tolist()))
    keys = sorted(seen)[::7]
    _compare_rows(e, o, np.array([k[0] for k in keys], np.uint64), np.array([k[1] for k in keys], np.uint32))
    e.close()


def test_vc_device_pointer_batches_match_oracle():
    dev = torch.device("cuda", 0)
    K, local = 3, 1
    rng = np.random.default_rng(31)
    e = bmx.EngineVC(1000, K, local); o = OracleVC(K, local)        # grows while device-pointer batches are in flight
    seen = set()
    for b in range(5):
        ids, fields, clocks, val = _rand_batch(rng, 7000, 20000, K, 4 + b, 3, hot=0.25)
        n = len(ids)
        t = [torch.from_numpy(x).to(dev) for x in (ids.view(np.int64), fields.view(np.int32), clocks.view(np.int32).reshape(-1), val)]
        upd = torch.zeros(n, dtype=torch.int32, device=dev); nu = torch.zeros(1, dtype=torch.int64, device=dev)
        fl = torch.zeros(n, dtype=torch.uint8, device=dev)
        e.merge_batch_dev(n, *t, updated=upd, n_updated=nu, flags=fl)
        e.sync()
        f2, u2 = o.merge_batch(ids, fields, clocks, val)
        k = int(nu.item())
        assert np.array_equal(fl.cpu().numpy(), f2), b
        assert np.array_equal(upd[:k].cpu().numpy().view(np.uint32), u2), b
        assert e.row_count() == len(o)
        seen.update(zip(ids.tolist(), fields.tolist()))
    keys = sorted(seen)[::5]
    _compare_rows(e, o, np.array([k[0] for k in keys], np.uint64), np.array([k[1] for k in keys], np.uint32))
    bad = torch.tensor([1 << 60], dtype=torch.int64, device=dev)
    one = [torch.from_numpy(x).to(dev) for x in (np.array([5], np.int64), np.array([7], np.int32), np.ones(K, np.int32))]
    e.merge_batch_dev(1, *one, bad)
    with pytest.raises(bmx.BmxError) as ei:
        e.sync()
    assert ei.value.code == bmx.ERR_RANGE
    e.close()


def test_vc_long_lists_are_linear_not_quadratic():
    """10^5 deltas on ONE key plus a field of medium lists (17..400 deltas per key: just over the selection cut-off and well beyond):
    the long-list path (k_vc_resolve_long) must give the reference's flags, updated index and final rows, and finish in bounded time
    (a quadratic selection over 10^5 nodes would take minutes to hours of dependent loads)."""
    import time
    K, local = 3, 1
    rng = np.random.default_rng(21)
    n_hot = 100_000
    keys_mid = np.array([keyid(1000 + i) for i in range(300)], np.uint64)
    mult = rng.integers(17, 400, len(keys_mid))
    ids = np.concatenate([np.full(n_hot, keyid(77), np.uint64), np.repeat(keys_mid, mult)])
    perm = rng.permutation(len(ids))
    ids = ids[perm]
    n = len(ids)
    fields = np.full(n, F0, np.uint32)
    clocks = rng.integers(0, 6, (n, K)).astype(np.uint32); val = rng.integers(-3, 4, n).astype(np.int64)
    e = bmx.EngineVC(4096, K, local); o = OracleVC(K, local)
    for b in range(2):     # second batch: the rows exist
        t0 = time.perf_counter()
        f1, u1 = e.merge_batch(ids, fields, clocks, val)
        dt = time.perf_counter() - t0
        f2, u2 = o.merge_batch(ids, fields, clocks, val)
        assert np.array_equal(f1, f2) and np.array_equal(u1, u2), b
        assert dt < 5.0, "long-list batch took %.1f s" % dt
        clocks = clocks + rng.integers(0, 2, (n, K)).astype(np.uint32)
    allk = np.concatenate([[keyid(77)], keys_mid]).astype(np.uint64)
    _compare_rows(e, o, allk, np.full(len(allk), F0, np.uint32))
    e.close()


def test_vc_scans_over_the_table_match_the_rows_read_back():
    """range / equals / count over the K-writer rows: the ids a scan returns are exactly the rows whose stored value (read back row by row) lies in
    the range, for every field, before and after the table grew."""
    rng = np.random.default_rng(77)
    K = 3
    e = bmx.EngineVC(2000, K, 1)                     # small: the table grows along the way
    seen = set()
    for b in range(5):
        ids, fields, clocks, val = _rand_batch(rng, 6000, 4000, K, 3 + b, 40, hot=0.1)
        e.merge_batch(ids, fields, clocks, val)
        seen.update(zip(ids.tolist(), fields.tolist()))
        keys = sorted(seen)
        kid = np.array([k[0] for k in keys], np.uint64); kf = np.array([k[1] for k in keys], np.uint32)
        _, vals, state = e.get_rows(kid, kf)
        for f in np.unique(kf):
            for lo, hi in [(-1000, 1000), (0, 0), (-5, 7), (10, -10)]:
                want = np.sort(kid[(kf == f) & (state != 0) & (vals >= lo) & (vals <= hi)])
                got = np.sort(e.scan_range(int(f), lo, hi))
                assert np.array_equal(got, want), (b, int(f), lo, hi, len(got), len(want))
                assert e.scan_range(int(f), lo, hi, count_only=True) == len(want)
    e.close()
