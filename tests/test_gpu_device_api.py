"""GPU: the asynchronous device-pointer side of the C ABI (torch tensors only carry the memory),
32-byte record input, owner partition, the caller-guaranteed unique-keys mode, per-kernel profiling."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import bmx
from oracle import streams
from bmx import synth
from oracle.oracle import Oracle, INSERT_REFERENCE, INSERT_DELTA, rows_digest, owner_of as o_owner

F0 = streams.field_hash(0)


def _dev(cols, dev):
    id, f, ts, val = cols
    return (torch.from_numpy(np.ascontiguousarray(id).view(np.int64)).to(dev), torch.from_numpy(np.ascontiguousarray(f).view(np.int32)).to(dev),
            torch.from_numpy(np.ascontiguousarray(ts)).to(dev), torch.from_numpy(np.ascontiguousarray(val)).to(dev))


def test_device_pointer_merge_matches_oracle():
    dev = torch.device("cuda", 0)
    R, D = 200_000, 50_000
    res = synth.big_resident(R, seed=31)
    o = Oracle(); o.load_rows(*res)
    with bmx.Engine(2 * (R + 4 * D)) as e:
        e.load_rows_dev(R, *_dev(res, dev))
        applied = torch.zeros(D, dtype=torch.int32, device=dev)
        n_applied = torch.zeros(1, dtype=torch.int64, device=dev)
        stats = torch.zeros(4, dtype=torch.int64, device=dev)
        flags = torch.zeros(D, dtype=torch.uint8, device=dev)
        for b in range(4):
            d = synth.big_deltas(D, R, seed=32, insert_pct=10, hot_pct=25, hot_keys=64, unique=False, batch=b)
            dd = _dev(d, dev)
            e.merge_batch_dev(D, *dd, INSERT_REFERENCE, applied=applied, n_applied=n_applied, flags=flags, stats=stats)
            e.sync()
            _, ow = o.merge_batch(*d)
            na = int(n_applied.item())
            assert na == len(ow) == int(stats[0].item())
            assert np.array_equal(applied[:na].cpu().numpy().view(np.uint32), ow)
            assert int(stats[2].item()) == len(o)
        assert rows_digest(*e.dump_rows()) == o.digest()


def test_unique_keys_mode_equals_default_on_unique_batches():
    dev = torch.device("cuda", 0)
    R, D = 300_000, 100_000
    res = synth.big_resident(R, seed=41)
    o = Oracle(); o.load_rows(*res)
    with bmx.Engine(2 * (R + 3 * D)) as e:
        e.load_rows(*res)
        for b in range(3):
            d = synth.big_deltas(D, R, seed=42, insert_pct=10, unique=True, batch=b, drift=60_000)
            assert len(np.unique(d[0])) == D
            applied, flags, st = e.merge_batch(*d, insert_mode=INSERT_REFERENCE | bmx.MERGE_UNIQUE_KEYS)
            of, ow = o.merge_batch(*d)
            assert np.array_equal(applied, ow) and np.array_equal(flags, of) and st.n_conflicts == 0
        assert rows_digest(*e.dump_rows()) == o.digest()


def test_partition_by_owner_is_stable_and_complete_then_merge_records():
    dev = torch.device("cuda", 0)
    n, G = 100_000, 8
    d = synth.big_deltas(n, 500_000, seed=51, insert_pct=20, hot_pct=20, hot_keys=100, unique=False)
    dd = _dev(d, dev)
    recs = torch.zeros((n, 4), dtype=torch.int64, device=dev)
    counts = torch.zeros(G, dtype=torch.int64, device=dev)
    with bmx.Engine(1_000_000) as e:
        e.partition_by_owner_dev(n, *dd, G, recs, counts)
        e.sync()
        c = counts.cpu().numpy()
        r = recs.cpu().numpy().view(bmx.DELTA_REC_DTYPE).reshape(-1)
        own = o_owner(d[0], G)
        assert c.tolist() == np.bincount(own, minlength=G).tolist() and c.sum() == n
        off = 0
        for g in range(G):
            seg = r[off:off + c[g]]
            src = np.nonzero(own == g)[0]                     # stable: original order inside each shard
            assert np.array_equal(seg["aux"], src.astype(np.uint32))
            assert np.array_equal(seg["id"], d[0][src]) and np.array_equal(seg["field"], d[1][src])
            assert np.array_equal(seg["ts"], d[2][src]) and np.array_equal(seg["val"], d[3][src])
            off += c[g]
        # records as merge input: same result as the column input
        o = Oracle()
        _, ow = o.merge_batch(r["id"], r["field"], r["ts"], r["val"])
        applied = torch.zeros(n, dtype=torch.int32, device=dev)
        n_applied = torch.zeros(1, dtype=torch.int64, device=dev)
        e.merge_records_dev(n, recs, INSERT_REFERENCE, applied=applied, n_applied=n_applied)
        e.sync()
        na = int(n_applied.item())
        assert np.array_equal(applied[:na].cpu().numpy().view(np.uint32), ow)
        assert rows_digest(*e.dump_rows()) == o.digest()


def test_padding_records_are_skipped():
    dev = torch.device("cuda", 0)
    r = np.zeros(6, dtype=bmx.DELTA_REC_DTYPE)
    r["id"] = [11, 2**64 - 1, 12, 2**64 - 1, 11, 2**64 - 1]
    r["field"] = F0; r["ts"] = [5, 0, 6, 0, 7, 0]; r["val"] = [1, 0, 2, 0, 3, 0]
    recs = torch.from_numpy(r.view(np.int64).reshape(6, 4)).to(dev)
    n_applied = torch.zeros(1, dtype=torch.int64, device=dev)
    applied = torch.zeros(6, dtype=torch.int32, device=dev)
    with bmx.Engine(100) as e:
        e.merge_records_dev(6, recs, INSERT_DELTA, applied=applied, n_applied=n_applied)
        e.sync()
        assert applied[:int(n_applied.item())].cpu().tolist() == [2, 4]
        assert e.row_count() == 2 and e.get_row(11, F0) == (7, 3)


def test_profile_hooks_report_three_stages():
    R, D = 100_000, 50_000
    with bmx.Engine(4 * R) as e:
        e.load_rows(*synth.big_resident(R, seed=61))
        e.profile_enable(True)
        for b in range(3):
            e.merge_batch(*synth.big_deltas(D, R, seed=62, batch=b), want_flags=False)
        ms, n = e.profile_read()
        assert n == 3 and all(v > 0 for v in ms.values())
        e.profile_enable(False)


def test_sharded_step_over_rccl_world1():
    """The product's sharded step (K7 partition -> all_to_all over RCCL -> merge_records) with a one-rank group:
    every code path of the N>1 bench except the cross-GPU links. Must equal the plain merge."""
    import os
    import torch.distributed as dist
    from bmx.sharded import ShardedGraph, EngineOps
    dev = torch.device("cuda", 0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        R, D = 200_000, 60_000
        o = Oracle(); o.load_rows(*synth.big_resident(R, seed=1, T0=1000, DT=1000))
        with bmx.Engine(2 * (R + 3 * D)) as e:
            sg = ShardedGraph(EngineOps(e, dev), dist, 0, 1)
            assert sg.load_owned_resident(R, T0=1000, DT=1000) == R
            for b in range(3):
                d = synth.big_deltas(D, R, seed=71, T0=1000, DT=1000, insert_pct=10, hot_pct=20, hot_keys=100, unique=False, batch=b)
                assert sg.merge_step(D, *_dev(d, dev)) == D
                applied, recv = sg.last_applied()
                _, ow = o.merge_batch(*d)
                # records keep their origin index in `aux`; with one shard the partition is the identity
                assert np.array_equal(applied.cpu().numpy().view(np.uint32), ow)
            assert rows_digest(*e.dump_rows()) == o.digest()
            # pipelined mode: slabs + second stream, two batches in flight
            sg.setup_pipeline(D, slack=1.05)
            ds = [synth.big_deltas(D, R, seed=72, T0=1000, DT=1000, insert_pct=10, hot_pct=20, hot_keys=100, unique=False, batch=b) for b in range(4)]
            dd = [_dev(d, dev) for d in ds]
            tk = sg.route(D, *dd[0])
            outs = []
            for b in range(4):
                nxt = sg.route(D, *dd[b + 1]) if b + 1 < 4 else None
                p = sg.merge(tk)
                sg.ops.sync()
                na = int(p["n_applied"].item())
                outs.append(p["applied"][:na].cpu().numpy().view(np.uint32).copy())
                tk = nxt
            assert not sg.overflowed()
            for b in range(4):
                _, ow = o.merge_batch(*ds[b])
                assert np.array_equal(outs[b], ow), b     # one shard: slab 0 holds the batch in order, indices match
            assert rows_digest(*e.dump_rows()) == o.digest()
            sg.setup_pipeline(D, slack=0.5)               # too small on purpose: overflow must be reported
            sg.merge(sg.route(D, *dd[0]))
            assert sg.overflowed()
            sg.ops.close()
            e.set_stream(None)
    finally:
        dist.destroy_process_group()



def test_sequence_words_order_two_streams():
    """bmx_seq_signal / bmx_seq_wait: a consumer stream sees the producer's data once the sequence word reaches the value."""
    dev = torch.device("cuda", 0)
    a, b = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    seq = torch.zeros(1, dtype=torch.int64, device=dev)
    n = 1 << 22
    x = torch.zeros(n, dtype=torch.int64, device=dev)
    y = torch.zeros(n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    with bmx.Engine(1000) as e:
        for k in range(1, 6):
            with torch.cuda.stream(a):
                x.fill_(k)                                   # producer work on stream a
            e.seq_signal(a.cuda_stream, seq, k)              # ... then seq = k
            e.seq_wait(b.cuda_stream, seq, k)                # consumer: wait on the device for seq >= k
            with torch.cuda.stream(b):
                y.copy_(x)
            b.synchronize()
            assert int(seq.item()) == k and bool((y == k).all())
            e.seq_wait(b.cuda_stream, seq, k - 1)            # already satisfied: returns at once
            b.synchronize()
        e.sync()
        with pytest.raises(bmx.BmxError):
            e.seq_wait(b.cuda_stream, None, 1)


def test_pipelined_host_batches_submit_collect():
    """bmx_merge_submit / bmx_merge_collect: two host batches in flight (upload of b+1 under the merge of b), results one batch late,
    same winners / flags / state as the synchronous calls; a third submit, an unknown ticket and an out-of-order collect are refused."""
    R, D, NBATCH = 50_000, 20_000, 6
    res = synth.big_resident(R, seed=31)
    bs = [synth.big_deltas(D, R, seed=32, insert_pct=10, hot_pct=20, hot_keys=64, unique=False, batch=b) for b in range(NBATCH)]
    o = Oracle(); o.load_rows(*res)
    want = [o.merge_batch(*b) for b in bs]
    with bmx.Engine(4 * (R + NBATCH * D)) as e:
        e.load_rows(*res)
        t_prev = e.merge_submit(*bs[0], want_flags=True)
        for b in range(1, NBATCH):
            t = e.merge_submit(*bs[b], want_flags=(b % 2 == 0))
            if b == 1:
                with pytest.raises(bmx.BmxError):            # a third batch before the first is collected
                    e.merge_submit(*bs[2])
                with pytest.raises(bmx.BmxError):            # out of order
                    e.merge_collect(t)
                with pytest.raises(bmx.BmxError):            # a synchronous merge while batches are in flight
                    e.merge_batch(*bs[0])
            applied, flags, st = e.merge_collect(t_prev)
            assert np.array_equal(applied, want[b - 1][1]), b
            assert st.n_applied == len(applied)
            t_prev = t
        applied, flags, st = e.merge_collect(t_prev)
        assert np.array_equal(applied, want[-1][1]) and st.n_rows == len(o)
        with pytest.raises(bmx.BmxError):
            e.merge_collect(t_prev)                          # already collected
        assert rows_digest(*e.dump_rows()) == o.digest()
        a2, _, _ = e.merge_batch(*bs[0])                     # the synchronous form still works afterwards
        _, w2 = o.merge_batch(*bs[0])
        assert np.array_equal(a2, w2)


def test_page_locked_host_columns_are_free_again_when_submit_returns():
    """bmx_host_alloc: ONE page-locked column set is refilled for every batch — right after bmx_merge_submit returns, while that batch is still
    in flight (copies from such memory are asynchronous, the call waits for its upload) — and also feeds the synchronous call and bmx_put_rows;
    winners and final rows equal the oracle's."""
    R, D, NBATCH = 60_000, 40_000, 5            # above the small-batch limit: the staging path
    res = synth.big_resident(R, seed=41)
    bs = [synth.big_deltas(D, R, seed=42, insert_pct=10, hot_pct=20, hot_keys=64, unique=False, batch=b) for b in range(NBATCH)]
    o = Oracle(); o.load_rows(*res)
    want = [o.merge_batch(*b)[1] for b in bs]
    hb, *pin = bmx.host_columns(D)
    assert all(a.flags["C_CONTIGUOUS"] and len(a) == D for a in pin)

    def fill(b):
        for dst, src in zip(pin, bs[b]):
            dst[:] = src

    with bmx.Engine(4 * (R + NBATCH * D)) as e:
        e.load_rows(*res)
        fill(0)
        t_prev = e.merge_submit(*pin)
        for b in range(1, NBATCH - 1):
            fill(b)                                           # overwrites what batch b-1 was uploaded from
            t = e.merge_submit(*pin)
            applied, _, _ = e.merge_collect(t_prev)
            assert np.array_equal(applied, want[b - 1]), b
            t_prev = t
        for a in pin:
            a[:] = 0                                          # and scribbled over while the last one is in flight
        applied, _, _ = e.merge_collect(t_prev)
        assert np.array_equal(applied, want[NBATCH - 2])
        fill(NBATCH - 1)
        applied, _, _ = e.merge_batch(*pin, want_flags=False)
        assert np.array_equal(applied, want[NBATCH - 1])
        assert rows_digest(*e.dump_rows()) == o.digest()
    hb.close()
    with pytest.raises(bmx.BmxError):
        bmx.HostBuffer(0)


@pytest.mark.parametrize("n", [1, 32767, 32768, 32769])
def test_host_batches_on_both_sides_of_the_small_batch_limit(n):
    """Host batches of up to 32768 deltas go through mapped host memory, larger ones through the staging copies: same winners, flags, stats,
    point reads and small scans either way."""
    from oracle.oracle import Oracle
    from bmx import synth
    rng = np.random.default_rng(n)
    res = synth.big_resident(50_000, seed=9)
    o = Oracle(); o.load_rows(*res)
    with bmx.Engine(400_000) as e:
        e.load_rows(*res)
        for b in range(3):
            d = synth.big_deltas(n, 50_000, seed=70, insert_pct=15, hot_pct=20, hot_keys=11, unique=False, batch=b)
            strict = bmx.MERGE_STRICT_FLAGS if b == 1 else 0
            applied, flags, st = e.merge_batch(*d, insert_mode=bmx.INSERT_REFERENCE | strict, want_flags=True)
            of, ow = o.merge_batch(*d)
            assert np.array_equal(applied, ow), (n, b)
            if strict:
                assert np.array_equal(flags, of), (n, b)
            assert st.n_applied == len(ow) and st.n_rows == len(o) == e.row_count()
        k = min(len(res[0]), 9000)                   # point reads: 8192 keys is the limit of the mapped-memory answer
        for m in (1, 8192, k):
            ts, val, found = e.get_rows(res[0][:m], res[1][:m])
            for i in (0, m - 1):
                assert found[i] and (int(ts[i]), int(val[i])) == o.get_row(int(res[0][i]), int(res[1][i]))
        f0 = int(res[1][0])
        for lo, hi in [(5, 5), (-(1 << 31), 1 << 31), (0, 1 << 24)]:   # a few ids, every id (more than the mapped answer holds), many
            assert np.array_equal(np.sort(e.scan_range(f0, lo, hi)), np.sort(o.scan_range(f0, lo, hi)))
            assert e.scan_count(f0, lo, hi) == o.scan_count(f0, lo, hi)
