"""GPU: deferred compaction (include/bmx.h): in a stream of device-resident batches the winner compaction of batch b runs on a second stream
under the probe kernel of batch b + 1. Every batch's winner list, counts and the final table must equal the oracle's sequential loop
(src/bullet-network-sync.js:551-569 -> src/bullet-crt.js:164-279) whatever is interleaved with the merges."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import bmx
from bmx import synth
from oracle import streams
from oracle.oracle import Oracle, INSERT_REFERENCE, rows_digest


def _dev(cols, dev):
    id, f, ts, val = cols
    return (torch.from_numpy(np.ascontiguousarray(id).view(np.int64)).to(dev), torch.from_numpy(np.ascontiguousarray(f).view(np.int32)).to(dev),
            torch.from_numpy(np.ascontiguousarray(ts)).to(dev), torch.from_numpy(np.ascontiguousarray(val)).to(dev))


def _check_batches(o, host_batches, applied, n_applied, stats=None):
    na = n_applied.cpu().numpy()
    for b, d in enumerate(host_batches):
        _, ow = o.merge_batch(*d)
        assert int(na[b]) == len(ow), (b, int(na[b]), len(ow))
        assert np.array_equal(applied[b][:len(ow)].cpu().numpy().view(np.uint32), ow), b
        if stats is not None:
            assert int(stats[b][0].item()) == len(ow) and int(stats[b][2].item()) == len(o), b


@pytest.mark.parametrize("hot", [0, 30])
def test_stream_of_device_batches_equals_oracle_and_runs_on_the_side_stream(hot):
    dev = torch.device("cuda", 0)
    R, D, NB = 400_000, 150_000, 9
    res = synth.big_resident(R, seed=71)
    o = Oracle(); o.load_rows(*res)
    hb = [synth.big_deltas(D, R, seed=72, insert_pct=10, hot_pct=hot, hot_keys=97, unique=(hot == 0), batch=b, drift=40_000) for b in range(NB)]
    db = [_dev(d, dev) for d in hb]
    applied = torch.zeros((NB, D), dtype=torch.int32, device=dev)
    n_applied = torch.zeros(NB, dtype=torch.int64, device=dev)
    stats = torch.zeros((NB, 4), dtype=torch.int64, device=dev)
    with bmx.Engine(2 * (R + NB * D)) as e:
        e.load_rows(*res)
        for b in range(NB):
            e.merge_batch_dev(D, *db[b], INSERT_REFERENCE, applied=applied[b], n_applied=n_applied[b:b + 1], stats=stats[b])
        e.sync()
        deferred, side = e.deferred_counts()
        assert deferred == NB and side == NB - 1, (deferred, side)       # the last one was launched by bmx_sync on the engine's own stream
        _check_batches(o, hb, applied, n_applied, stats)
        assert e.row_count() == len(o) and rows_digest(*e.dump_rows()) == o.digest()


def test_switched_off_is_the_same_answer_and_defers_nothing():
    dev = torch.device("cuda", 0)
    R, D, NB = 200_000, 100_000, 4
    res = synth.big_resident(R, seed=73)
    o = Oracle(); o.load_rows(*res)
    hb = [synth.big_deltas(D, R, seed=74, insert_pct=10, hot_pct=20, hot_keys=50, unique=False, batch=b, drift=40_000) for b in range(NB)]
    db = [_dev(d, dev) for d in hb]
    applied = torch.zeros((NB, D), dtype=torch.int32, device=dev)
    n_applied = torch.zeros(NB, dtype=torch.int64, device=dev)
    with bmx.Engine(2 * (R + NB * D)) as e:
        e.set_deferred(False)
        e.load_rows(*res)
        for b in range(NB):
            e.merge_batch_dev(D, *db[b], INSERT_REFERENCE, applied=applied[b], n_applied=n_applied[b:b + 1])
        e.sync()
        assert e.deferred_counts() == (0, 0)
        _check_batches(o, hb, applied, n_applied)
        assert rows_digest(*e.dump_rows()) == o.digest()


def test_anything_between_two_merges_sees_the_finished_batch():
    """scans over a maintained index (the change log is written by the compaction), point reads, row counts, host batches, strict and
    unique-keys batches, small batches, a growth of the workspace: each is interleaved with deferring merges and compared with the oracle"""
    dev = torch.device("cuda", 0)
    R, D = 300_000, 120_000
    F = streams.field_hash(0)
    res = synth.big_resident(R, seed=75)
    o = Oracle(); o.load_rows(*res)
    n_applied = torch.zeros(1, dtype=torch.int64, device=dev)
    rng = np.random.default_rng(5)
    with bmx.Engine(2 * (R + 40 * D)) as e:
        e.load_rows(*res)
        e.index_build(F)
        for b in range(16):
            n = D if b != 9 else 2 * D + 777               # b == 9: the workspace grows while a compaction is only recorded
            d = synth.big_deltas(n, R, seed=76, insert_pct=10, hot_pct=15, hot_keys=40, unique=False, batch=b, drift=30_000)
            dd = _dev(d, dev)
            applied = torch.zeros(n, dtype=torch.int32, device=dev)
            e.merge_batch_dev(n, *dd, INSERT_REFERENCE, applied=applied, n_applied=n_applied)
            _, ow = o.merge_batch(*d)
            what = b % 8
            if what == 0:
                lo, hi = -(1 << 27), 1 << 27
                assert np.array_equal(np.sort(e.scan_range(F, lo, hi)), np.sort(o.scan_range(F, lo, hi)))
            elif what == 1:
                k = rng.integers(0, len(d[0]), 500)
                ts, val, found = e.get_rows(d[0][k], d[1][k])
                for i, kk in enumerate(k):
                    got = o.get_row(int(d[0][kk]), int(d[1][kk]))
                    assert found[i] and got == (int(ts[i]), int(val[i]))
            elif what == 2:
                assert e.row_count() == len(o)
            elif what == 3:
                d2 = synth.big_deltas(70_000, R, seed=77, insert_pct=10, hot_pct=15, hot_keys=40, unique=False, batch=b, drift=30_000)
                a2, _, st2 = e.merge_batch(*d2)
                _, ow2 = o.merge_batch(*d2)
                assert np.array_equal(a2, ow2) and st2.n_rows == len(o)
            elif what == 4:
                d2 = synth.big_deltas(90_000, R, seed=78, insert_pct=10, hot_pct=15, hot_keys=40, unique=False, batch=b, drift=30_000)
                a2, f2, _ = e.merge_batch(*d2, insert_mode=INSERT_REFERENCE | bmx.MERGE_STRICT_FLAGS)
                of2, ow2 = o.merge_batch(*d2)
                assert np.array_equal(a2, ow2) and np.array_equal(f2, of2)
            elif what == 5:
                d2 = synth.big_deltas(80_000, R, seed=79, insert_pct=10, unique=True, batch=b, drift=30_000)
                dd2 = _dev(d2, dev)
                ap2 = torch.zeros(80_000, dtype=torch.int32, device=dev)
                e.merge_batch_dev(80_000, *dd2, INSERT_REFERENCE | bmx.MERGE_UNIQUE_KEYS, applied=ap2, n_applied=n_applied)
                e.sync()
                _, ow2 = o.merge_batch(*d2)
                assert np.array_equal(ap2[:len(ow2)].cpu().numpy().view(np.uint32), ow2)
            elif what == 6:
                d2 = synth.big_deltas(3000, R, seed=80, insert_pct=10, hot_pct=15, hot_keys=40, unique=False, batch=b, drift=30_000)   # below the deferral threshold
                dd2 = _dev(d2, dev)
                ap2 = torch.zeros(3000, dtype=torch.int32, device=dev)
                e.merge_batch_dev(3000, *dd2, INSERT_REFERENCE, applied=ap2, n_applied=n_applied)
                e.sync()
                _, ow2 = o.merge_batch(*d2)
                assert np.array_equal(ap2[:len(ow2)].cpu().numpy().view(np.uint32), ow2)
            elif what == 7 and b == 7:                     # the table grows (every row re-inserted into a new table) while a compaction is only recorded
                e.reserve(4 * (R + 40 * D))
            elif what == 7:                                # rows decided elsewhere, stored as given, and the index dropped and built again
                k = rng.integers(0, R, 2000)
                pid, pf = res[0][k], res[1][k]
                pts = np.full(len(k), 9_000_000 + b, dtype=np.int64); pv = rng.integers(-1000, 1000, len(k)).astype(np.int64)
                _, first = np.unique(pid, return_index=True)     # one row per key
                e.put_rows(pid[first], pf[first], pts[first], pv[first]); o.put_rows(pid[first], pf[first], pts[first], pv[first])
                e.index_drop(F); e.index_build(F)
            e.sync()
            na = int(n_applied.item()) if what not in (5, 6) else len(ow)
            assert np.array_equal(applied[:len(ow)].cpu().numpy().view(np.uint32), ow), b
            if what not in (5, 6):
                assert na == len(ow)
        full, inc = e.index_refresh_counts()
        assert rows_digest(*e.dump_rows()) == o.digest()
        assert np.array_equal(np.sort(e.scan_range(F, -(1 << 40), 1 << 40)), np.sort(o.scan_range(F, -(1 << 40), 1 << 40)))
        assert e.deferred_counts()[0] >= 16


def test_inputs_may_be_overwritten_in_stream_order_and_the_fence_orders_the_outputs():
    """the engine runs on the CALLER's stream: the batch columns are overwritten by the caller's next kernel on that stream (an index is being
    maintained, so the compaction wants the deltas' fields — from its own copy), and the outputs are read on that stream behind bmx_merge_fence"""
    dev = torch.device("cuda", 0)
    R, D, NB = 250_000, 131_072, 6
    F = streams.field_hash(0)
    res = synth.big_resident(R, seed=81)
    o = Oracle(); o.load_rows(*res)
    hb = [synth.big_deltas(D, R, seed=82, insert_pct=10, hot_pct=10, hot_keys=30, unique=False, batch=b, drift=30_000) for b in range(NB)]
    st = torch.cuda.Stream(device=dev)
    with bmx.Engine(2 * (R + NB * D)) as e:
        e.load_rows(*res)
        e.index_build(F)
        e.sync()
        e.set_stream(st.cuda_stream)
        copies = []
        with torch.cuda.stream(st):
            src = [_dev(d, dev) for d in hb]
            work = [torch.empty_like(t) for t in src[0]]
            applied = torch.zeros((NB, D), dtype=torch.int32, device=dev)
            n_applied = torch.zeros(NB, dtype=torch.int64, device=dev)
            for b in range(NB):
                for w, s_ in zip(work, src[b]):
                    w.copy_(s_)                                     # the caller's ONE set of batch columns, refilled per batch on its stream
                e.merge_batch_dev(D, *work, INSERT_REFERENCE, applied=applied[b], n_applied=n_applied[b:b + 1])
                if b == 2:
                    e.merge_fence()
                    copies.append((b, applied[b].clone(), n_applied[b:b + 1].clone()))   # read on the stream, no host sync
            e.merge_fence()
            final_counts = n_applied.clone()
        st.synchronize()
        e.set_stream(None)
        _check_batches(o, hb, applied, final_counts)
        b, ap, na = copies[0]
        assert int(na.item()) == int(final_counts[b].item()) and torch.equal(ap, applied[b])
        assert rows_digest(*e.dump_rows()) == o.digest()
        assert np.array_equal(np.sort(e.scan_range(F, -(1 << 40), 1 << 40)), np.sort(o.scan_range(F, -(1 << 40), 1 << 40)))


def test_a_signal_on_the_contexts_stream_publishes_the_merge_outputs():
    """bmx_seq_signal on the context's own stream says "everything before this is done". With the deferral on, the last merge's compaction is
    only RECORDED when the call returns: the signal must order it in front of itself, so that a consumer stream woken by the word reads this batch's
    n_applied / applied_idx / stats, not stale ones. (Before the fix the word was set while k_compact_winners had not even been launched.)"""
    dev = torch.device("cuda", 0)
    R, D, NB = 250_000, 131_072, 4
    res = synth.big_resident(R, seed=91)
    o = Oracle(); o.load_rows(*res)
    hb = [synth.big_deltas(D, R, seed=92, insert_pct=10, hot_pct=10, hot_keys=30, unique=False, batch=b, drift=30_000) for b in range(NB)]
    consumer = torch.cuda.Stream(device=dev)
    with bmx.Engine(2 * (R + NB * D)) as e:
        e.load_rows(*res)
        db = [_dev(d, dev) for d in hb]
        applied = torch.zeros((NB, D), dtype=torch.int32, device=dev)
        n_applied = torch.full((NB,), -1, dtype=torch.int64, device=dev)
        stats = torch.zeros((NB, 8), dtype=torch.int64, device=dev)
        seen_n = torch.full((NB,), -7, dtype=torch.int64, device=dev)
        seen_applied = torch.zeros((NB, D), dtype=torch.int32, device=dev)
        word = torch.zeros(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        d0, _ = e.deferred_counts()
        for b in range(NB):
            e.merge_batch_dev(D, *db[b], INSERT_REFERENCE, applied=applied[b], n_applied=n_applied[b:b + 1], stats=stats[b])
            e.seq_signal(0, word, b + 1)                               # on the engine's own stream
            e.seq_wait(consumer.cuda_stream, word, b + 1)              # the consumer wakes on the word alone: no host sync, no fence
            with torch.cuda.stream(consumer):
                seen_n[b:b + 1].copy_(n_applied[b:b + 1])
                seen_applied[b].copy_(applied[b])
        consumer.synchronize()
        e.sync()
        assert e.deferred_counts()[0] - d0 == NB                       # every merge WAS deferred: the signal, not a switched-off deferral, ordered the outputs
        _check_batches(o, hb, seen_applied, seen_n)
        assert torch.equal(seen_n, n_applied)
        assert rows_digest(*e.dump_rows()) == o.digest()


def test_undocumented_insert_mode_bits_are_refused():
    dev = torch.device("cuda", 0)
    d = synth.big_deltas(1000, 5000, seed=3, insert_pct=10, unique=True)
    dd = _dev(d, dev)
    with bmx.Engine(20_000) as e:
        for bad in (0x4000, 0x800, 0x2, 0x10000 | bmx.MERGE_UNIQUE_KEYS):
            with pytest.raises(bmx.BmxError) as ei:
                e.merge_batch(*d, insert_mode=bad)
            assert ei.value.code == bmx.ERR_INVALID and "insert_mode" in str(ei.value)
            with pytest.raises(bmx.BmxError) as ei:
                e.merge_batch_dev(1000, *dd, bad)
            assert ei.value.code == bmx.ERR_INVALID
        assert e.row_count() == 0
        a, _, st = e.merge_batch(*d)          # the context is still usable
        assert st.n_applied == len(a) == 1000


def test_selfcheck_sees_no_torn_pair_and_its_control_does():
    reads, torn, control = bmx.selfcheck(0)
    assert reads > 10_000_000 and torn == 0
    assert control > 0, "the split-store control must tear, otherwise the check cannot see a tear"


def test_host_batches_work_when_the_page_locked_buffers_cannot_be_had():
    """a failed page-locked allocation used to free the staging tails that submit/collect keep using"""
    R, D = 50_000, 20_000
    res = synth.big_resident(R, seed=91)
    o = Oracle(); o.load_rows(*res)
    os.environ["BMX_TEST_FAIL_PINNED"] = "1"
    try:
        with bmx.Engine(4 * (R + 4 * D)) as e:
            e.load_rows(*res)
            for b, n in enumerate((D, 500, 40_000)):          # small (would take the mapped path), tiny, large (staged path)
                d = synth.big_deltas(n, R, seed=92, insert_pct=10, hot_pct=10, hot_keys=20, unique=False, batch=b)
                a, f, st = e.merge_batch(*d)
                of, ow = o.merge_batch(*d)
                assert np.array_equal(a, ow) and st.n_rows == len(o)
            ts, val, found = e.get_rows(res[0][:100], res[1][:100])
            assert found.all()
            assert rows_digest(*e.dump_rows()) == o.digest()
    finally:
        del os.environ["BMX_TEST_FAIL_PINNED"]


def test_a_context_destroyed_with_a_compaction_pending_finishes_it_first():
    """bmx_destroy right behind a deferring merge: the recorded compaction still writes the caller's winner list and count (they are the caller's
    memory, not the context's), and nothing of the context is freed underneath a kernel"""
    dev = torch.device("cuda", 0)
    R, D = 200_000, 100_000
    res = synth.big_resident(R, seed=91)
    o = Oracle(); o.load_rows(*res)
    d = synth.big_deltas(D, R, seed=92, insert_pct=10, hot_pct=10, hot_keys=30, unique=False, batch=0, drift=30_000)
    dd = _dev(d, dev)
    applied = torch.zeros(D, dtype=torch.int32, device=dev)
    n_applied = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    e = bmx.Engine(2 * (R + D))
    e.load_rows(*res)
    e.merge_batch_dev(D, *dd, INSERT_REFERENCE, applied=applied, n_applied=n_applied)
    assert e.deferred_counts()[0] == 1
    e.close()
    torch.cuda.synchronize()
    _, ow = o.merge_batch(*d)
    assert int(n_applied.item()) == len(ow)
    assert np.array_equal(applied[:len(ow)].cpu().numpy().view(np.uint32), ow)


@pytest.mark.parametrize("defer", [True, False])
def test_a_caller_far_ahead_of_the_device_waits_for_row_reports_and_neither_drains_nor_grows(defer):
    """a table with head room for TWO batches' worth of new rows and a stream of 24 batches that create none: the capacity guard counts every delta in
    flight as a possible new row, so from the third call on it has to wait for the oldest batch in flight to report its row count (the compaction
    writes the host-visible mirror) — with the deferral that report may belong to a compaction that is only recorded. Same answers, same table."""
    dev = torch.device("cuda", 0)
    R, D, NB = 400_000, 150_000, 24
    res = synth.big_resident(R, seed=93)
    o = Oracle(); o.load_rows(*res)
    hb = [synth.big_deltas(D, R, seed=94, insert_pct=0, hot_pct=20, hot_keys=64, unique=False, batch=b, drift=40_000) for b in range(NB)]
    db = [_dev(d, dev) for d in hb]
    applied = torch.zeros((NB, D), dtype=torch.int32, device=dev)
    n_applied = torch.zeros(NB, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    with bmx.Engine(R + 2 * D) as e:
        e.set_deferred(defer)
        e.load_rows(*res)
        slots0 = e.info().n_slots
        for b in range(NB):
            e.merge_batch_dev(D, *db[b], INSERT_REFERENCE, applied=applied[b], n_applied=n_applied[b:b + 1])
        e.sync()
        assert e.info().n_slots == slots0, "the table did not have to grow: no batch created a row"
        _check_batches(o, hb, applied, n_applied)
        assert e.row_count() == len(o) == R and rows_digest(*e.dump_rows()) == o.digest()
