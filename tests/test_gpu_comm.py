"""GPU: N shards in ONE process behind the C ABI (bmx_comm_*), N = 2, 4, 8 as logical shards on the one GPU of the box.
The union of the shards must equal a single unsharded merge bit for bit: winners (ascending indices into the caller's batch), row
count, state digest, point reads and scans are compared with the oracle, for the host-batch path and for the device-resident path
(owner partition scattering straight into the owners' receive slabs)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import bmx
from bmx import synth
from oracle.oracle import Oracle, INSERT_REFERENCE, rows_digest

R, D = 200_000, 60_000


def _dev(cols):
    i, f, t, v = cols
    d = torch.device("cuda", 0)
    return (torch.from_numpy(i.view(np.int64)).to(d), torch.from_numpy(f.view(np.int32)).to(d), torch.from_numpy(t).to(d), torch.from_numpy(v).to(d))


@pytest.mark.parametrize("N", [1, 2, 4, 8])
def test_host_batches_through_n_shards_equal_one_merge(N):
    res = synth.big_resident(R, seed=3, F=2)
    o = Oracle(); o.load_rows(*res)
    with bmx.Comm([0] * N, capacity_rows_per_shard=2 * (R + 4 * D) // N + 4096) as c:
        c.load_rows(*res)
        assert c.row_count() == len(o)
        for b in range(4):
            d = synth.big_deltas(D, R, seed=40 + N, F=2, insert_pct=10, hot_pct=25, hot_keys=200, unique=False, batch=b)
            applied, st = c.merge(*d)
            _, ow = o.merge_batch(*d)
            assert np.array_equal(applied, ow), (N, b, len(applied), len(ow))
            assert st.n_rows == len(o) and st.n_applied == len(ow)
        assert c.row_count() == len(o)
        assert rows_digest(*c.dump_rows()) == o.digest()
        # every shard holds exactly the rows it owns
        for g in range(N):
            e = bmx.Engine.__new__(bmx.Engine); e.L = c.L; e.h = C_ptr(c.L.bmx_comm_shard(c.h, g))
            gid = e.dump_rows()[0]
            assert (synth.owner_of_np(gid, N) == g).all()
            e.h = None
        # point reads routed by owner
        ids = np.concatenate([res[0][:500], synth.splitmix64_np(np.arange(10**9, 10**9 + 50, dtype=np.uint64))])
        flds = np.concatenate([res[1][:500], np.full(50, res[1][0], np.uint32)])
        ts, val, found = c.get_rows(ids, flds)
        for k in range(len(ids)):
            want = o.get_row(int(ids[k]), int(flds[k]))
            assert (want is None and not found[k]) or (want == (int(ts[k]), int(val[k])) and found[k])
        # sharded scan == scan of the whole graph
        f0 = int(res[1][0])
        for lo, hi in [(-(1 << 31), 1 << 31), (0, 1 << 20), (5, 5), (10, 3)]:
            got = np.sort(c.scan_range(f0, lo, hi)); want = np.sort(o.scan_range(f0, lo, hi))
            assert np.array_equal(got, want), (N, lo, hi)
            assert c.scan_count(f0, lo, hi) == len(want)
        got = np.sort(c.scan_filter([(f0, 0, 1 << 30), (int(res[1][1]), -(1 << 30), 0)]))
        want = np.sort(o.scan_filter_and([(f0, 0, 1 << 30), (int(res[1][1]), -(1 << 30), 0)]))
        assert np.array_equal(got, want)


def C_ptr(v):
    import ctypes
    return ctypes.c_void_p(v)


@pytest.mark.parametrize("N", [2, 4, 8])
def test_device_resident_batches_scatter_into_peer_slabs(N):
    """Every shard originates its own device-resident batch; owner partitions write straight into the owners' receive slabs."""
    res = synth.big_resident(R, seed=5)
    o = Oracle(); o.load_rows(*res)
    with bmx.Comm([0] * N, capacity_rows_per_shard=2 * (R + 4 * D) // N + 4096) as c:
        c.load_rows(*res)
        for step in range(3):
            host = [synth.big_deltas(D // N, R, seed=60 + i, insert_pct=10, hot_pct=20, hot_keys=100, unique=False, batch=step) for i in range(N)]
            devb = [(len(h[0]),) + _dev(h) for h in host]
            c.merge_dev(devb, slab_records=0)
            c.sync()
            for h in host:                 # a shard merges origin 0's run, then origin 1's, ...: the oracle applies them in that order
                o.merge_batch(*h)
            assert c.row_count() == len(o), (N, step)
        assert rows_digest(*c.dump_rows()) == o.digest()
        # too small a slab: records are dropped, the step reports it (sticky error), and re-sending through the host path repairs it
        # (hits only: the lexmax of a row is idempotent under re-delivery; a re-delivered INSERT would meet the row its first delivery
        # created with ts := 2 and replace that clock, exactly as the reference does when a sync chunk arrives twice)
        host = [synth.big_deltas(D // N, R, seed=90 + i, insert_pct=0, unique=False, batch=7) for i in range(N)]
        devb = [(len(h[0]),) + _dev(h) for h in host]
        c.merge_dev(devb, slab_records=max(1, D // N // N // 2))
        with pytest.raises(bmx.BmxError) as ei:
            c.sync()
        assert ei.value.code == bmx.ERR_OVERFLOW
        for h in host:
            c.merge(*h); o.merge_batch(*h)
        assert rows_digest(*c.dump_rows()) == o.digest()


def test_shards_grow_and_keep_indexes_fresh():
    """Tiny initial shards: every shard rehashes into larger tables as batches arrive (host and device path), indexes built before the
    growth are rebuilt on the next scan, and the union still equals the oracle."""
    N = 4
    o = Oracle()
    with bmx.Comm([0] * N, capacity_rows_per_shard=2048) as c:
        f0 = int(synth.field_hash(0))
        c.index_build(f0)                                   # on empty shards
        for b in range(5):
            d = synth.big_deltas(40_000, 30_000, seed=77, insert_pct=50, hot_pct=10, hot_keys=50, unique=False, batch=b)
            if b % 2 == 0:
                applied, st = c.merge(*d)
                _, ow = o.merge_batch(*d)
                assert np.array_equal(applied, ow), b
            else:
                parts = [tuple(x[i::N] for x in d) for i in range(N)]          # every shard originates a quarter
                c.merge_dev([(len(p[0]),) + _dev(tuple(np.ascontiguousarray(x) for x in p)) for p in parts], slab_records=20_000)
                c.sync()
                for p in parts:
                    o.merge_batch(*p)
            assert c.row_count() == len(o), b
            got = np.sort(c.scan_range(f0, -(1 << 40), 1 << 40)); want = np.sort(o.scan_range(f0, -(1 << 40), 1 << 40))
            assert np.array_equal(got, want), b
        assert rows_digest(*c.dump_rows()) == o.digest()


@pytest.mark.parametrize("N", [1, 3])
def test_value_ordered_views_on_every_shard(N):
    """bmx_comm_index_set_ordered: every shard answers from its own sorted copy of the index; the union equals the oracle's scan before and after merges
    that change the field (round 4: the queries right after a merge scanned the columns and a later one sorted again; since round 5 every shard patches its view from
    its own change log)."""
    o = Oracle()
    with bmx.Comm([0] * N, capacity_rows_per_shard=200_000) as c:
        f0 = int(synth.field_hash(0))
        res = synth.big_resident(120_000, seed=5)
        c.merge(*res); o.merge_batch(*res)
        c.index_build(f0)
        c.index_set_ordered(f0, 2)
        for b in range(3):
            for lo, hi in [(-(1 << 28), 1 << 28), (0, 1 << 30), (-(1 << 40), 1 << 40), (5, 4), (12345, 12345)] * 2:
                got = np.sort(c.scan_range(f0, lo, hi)); want = np.sort(o.scan_range(f0, lo, hi))
                assert np.array_equal(got, want), (b, lo, hi)
                assert c.scan_count(f0, lo, hi) == len(want)
            d = synth.big_deltas(30_000, 120_000, seed=6, insert_pct=20, hot_pct=10, hot_keys=50, unique=False, batch=b)
            applied, st = c.merge(*d)
            _, ow = o.merge_batch(*d)
            assert np.array_equal(applied, ow), b


@pytest.mark.parametrize("N", [1, 3, 8])
def test_small_host_batches_are_routed_on_the_host(N):
    """Batches of up to 32768 deltas take the host-routed path (owner per delta on the CPU, each shard's small-batch merge): same winners, in the
    caller's index space, as one sequential merge — including duplicates of one key inside a batch and keys created by the batch."""
    rng = np.random.default_rng(50 + N)
    res = synth.big_resident(20_000, seed=5, F=2)
    o = Oracle(); o.load_rows(*res)
    with bmx.Comm([0] * N, capacity_rows_per_shard=200_000) as c:
        c.load_rows(*res)
        for b, n in enumerate([1, 2, 50, 777, 5000, 32768, 1]):
            d = synth.big_deltas(n, 20_000, seed=60 + N, F=2, insert_pct=20, hot_pct=30, hot_keys=7, unique=False, batch=b)
            applied, st = c.merge(*d)
            _, ow = o.merge_batch(*d)
            assert np.array_equal(applied, ow), (N, b, n)
            assert st.n_rows == len(o) and st.n_applied == len(ow), (N, b, n)
        assert rows_digest(*c.dump_rows()) == o.digest()


@pytest.mark.parametrize("nshards", [1, 2, 4])
def test_put_rows_and_tombstones_over_shards(nshards):
    """bmx_comm_put_rows: rows decided on the host reach the shard that owns their node; tombstoned rows leave every shard's scans and dumps."""
    from oracle.oracle import VAL_DELETED
    from oracle import streams
    rng = np.random.default_rng(90 + nshards)
    n = 30_000
    ids = streams.splitmix64_np(np.arange(1, n + 1, dtype=np.uint64))
    f = streams.field_hash(1)
    o = Oracle()
    with bmx.Comm([0] * nshards, capacity_rows_per_shard=n) as c:
        v = rng.integers(-100, 101, n).astype(np.int64)
        c.load_rows(ids, np.full(n, f, np.uint32), np.full(n, 50, np.int64), v); o.load_rows(ids, np.full(n, f, np.uint32), np.full(n, 50, np.int64), v)
        c.index_build(f)
        sel = rng.choice(n, n // 3, replace=False)
        vals = rng.integers(-100, 101, len(sel)).astype(np.int64)
        vals[::4] = VAL_DELETED
        d = (ids[sel], np.full(len(sel), f, np.uint32), rng.integers(1, 90, len(sel)).astype(np.int64), vals)
        c.put_rows(*d); o.put_rows(*d)
        for lo, hi in [(-2**62, 2**62), (-10, 10), (0, 0)]:
            assert np.array_equal(np.sort(c.scan_range(f, lo, hi)), np.sort(o.scan_range(f, lo, hi))), (lo, hi)
        assert rows_digest(*c.dump_rows()) == o.digest()
    o.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two physical GPUs (the builder's box has one): peer access, cross-device events and peer stores run here first")
@pytest.mark.parametrize("ndev", [2, 4, 8])
def test_physical_devices_equal_one_merge(ndev):
    """The same checks as the logical-shard tests with every shard on its OWN GPU: host batches (device-to-device run copies, the shared winner
    byte map on shard 0 written by peers), device batches (owner partition storing straight into peer memory), sharded scans, puts."""
    if torch.cuda.device_count() < ndev:
        pytest.skip("needs %d GPUs" % ndev)
    res = synth.big_resident(R, seed=3, F=2)
    o = Oracle(); o.load_rows(*res)
    with bmx.Comm(list(range(ndev)), capacity_rows_per_shard=2 * (R + 4 * D) // ndev + 4096) as c:
        c.load_rows(*res)
        assert c.row_count() == len(o)
        for b in range(3):
            d = synth.big_deltas(D, R, seed=40 + b, insert_pct=10, hot_pct=20, hot_keys=50, unique=False, batch=b)
            applied, st = c.merge(*d)
            _, want = o.merge_batch(*d)
            assert np.array_equal(applied, want), b
        assert rows_digest(*c.dump_rows()) == o.digest()
        f = int(res[1][0])
        lo, hi = -(1 << 28), 1 << 28
        assert np.array_equal(np.sort(c.scan_range(f, lo, hi)), np.sort(o.scan_range(f, lo, hi)))
    o.close()


@pytest.mark.parametrize("N", [2, 8])
def test_host_batch_with_every_delta_on_one_shard(N):
    """A host batch whose node ids all belong to ONE shard: the slab sized for a uniform owner hash is too small, the batch is partitioned once more
    into slabs of the largest run before anything is merged. Winners and state as one unsharded merge."""
    from oracle import streams
    rng = np.random.default_rng(5 + N)
    cand = streams.splitmix64_np(np.arange(1, 2_000_000, dtype=np.uint64))
    mine = cand[bmx.owner_of(cand, N) == N - 1][:90_000]
    assert len(mine) == 90_000
    f = streams.field_hash(1)
    o = Oracle()
    with bmx.Comm([0] * N, capacity_rows_per_shard=400_000) as c:
        for b in range(2):
            ids = mine[rng.integers(0, len(mine), 70_000)]
            d = (ids, np.full(len(ids), f, np.uint32), rng.integers(1, 50, len(ids)).astype(np.int64), rng.integers(-9, 10, len(ids)).astype(np.int64))
            applied, st = c.merge(*d)
            _, want = o.merge_batch(*d)
            assert np.array_equal(applied, want), b
        assert c.row_count() == len(o)
        assert rows_digest(*c.dump_rows()) == o.digest()
    o.close()
