"""GPU: value-ordered view of an index (include/bmx.h bmx_index_set_ordered). The reference's index is a Map keyed by value (src/bullet-query.js:30-73):
equals() is one lookup (:186-210), range() walks distinct values (:221-261). With the view the device answers the same queries by two searches on a
sorted copy of the column and one contiguous copy; the SET of matches must equal a ground-truth scan of the same rows whatever path answered, positions
and ids must name the same rows in the same order, and the view must notice every change of the field and nothing else."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import bmx
from oracle import streams
from oracle.oracle import VAL_DELETED

FA, FS, FO = streams.fnv1a32("age"), streams.fnv1a32("score"), streams.fnv1a32("other")


def _want(ids, vals, lo, hi, alive=None):
    m = (vals >= lo) & (vals <= hi)
    if alive is not None:
        m &= alive
    return np.sort(ids[m])


def _q(e, f, ids, vals, lo, hi, alive=None):
    """ONE query (the re-sort policy counts queries), compared as a set"""
    want = _want(ids, vals, lo, hi, alive)
    got = e.scan_range(f, lo, hi)
    assert len(got) == len(want) and np.array_equal(np.sort(got), want), (lo, hi, len(got), len(want))


def _check(e, f, ids, vals, lo, hi, alive=None):
    want = _want(ids, vals, lo, hi, alive)
    got = e.scan_range(f, lo, hi)
    assert len(got) == len(want) and np.array_equal(np.sort(got), want), (lo, hi, len(got), len(want))
    assert e.scan_count(f, lo, hi) == len(want)
    pos = e.scan_range_pos(f, lo, hi)
    col = e.index_ids(f)
    # positions and ids name the same rows. (Not necessarily in the same order in two separate calls: a view with a pending patch lists main's survivors, then the
    # inserted keys, and the rewrite of main — which runs behind an answer — turns that into one (value, position) run between the two calls.)
    assert len(pos) == len(want) and np.array_equal(np.sort(col[pos]), np.sort(got)), (lo, hi, "positions and ids name the same rows")
    return got


@pytest.mark.parametrize("wide", [False, True])
def test_queries_through_the_view_equal_a_scan(wide):
    R = 3_000_017
    f = FS if wide else FA
    sh = 33 if wide else 0
    ids = streams.splitmix64_np(np.arange(1, R + 1, dtype=np.uint64))
    with np.errstate(over="ignore"):
        vals = ((streams.splitmix64_np(ids ^ np.uint64(0x5151)) % np.uint64(5000)).astype(np.int64) - 2500) << sh
    with bmx.Engine(R + 1000) as e:
        e.load_rows(ids, np.full(R, f, np.uint32), np.full(R, 5, np.int64), vals)
        e.index_build(f)
        before = {(lo, hi): np.sort(e.scan_range(f, lo << sh, hi << sh)) for lo, hi in [(7, 7), (-100, 100)]}      # answered by the column scan
        e.index_set_ordered(f, 1)
        assert e.index_ordered_info(f) == (1, False, 0)
        rng = np.random.default_rng(3)
        qs = [(7, 7), (-100, 100), (-2500, 2499), (-2500, -2500), (2499, 2499), (2500, 9000), (-9000, -2501), (10, 9), (0, 0), (-1, 1), (-(1 << 20), 1 << 20)]
        qs += [tuple(sorted(rng.integers(-2600, 2600, 2).tolist())) for _ in range(60)]
        for lo, hi in qs:
            got = _check(e, f, ids, vals, lo << sh, hi << sh)
            if (lo, hi) in before:
                assert np.array_equal(np.sort(got), before[(lo, hi)])
        a, valid, sorts = e.index_ordered_info(f)
        assert valid and sorts == 1, "one sort served every query: nothing was written in between"
        # (value, position) order: values ascend, positions ascend inside one value
        pos = e.scan_range_pos(f, -50 << sh, 50 << sh)
        col = e.index_ids(f)
        order = np.argsort(ids, kind="stable"); vv = vals[order][np.searchsorted(ids[order], col[pos])]
        assert np.all(np.diff(vv) >= 0) and np.all((np.diff(vv) > 0) | (np.diff(pos.astype(np.int64)) > 0))
        # truncated answers keep the full count; a device buffer gets the same answer
        got = e.scan_range(f, -100 << sh, 100 << sh, cap=777); full = e.scan_range(f, -100 << sh, 100 << sh)
        assert len(got) == 777 and np.array_equal(got, full[:777])
        dev = torch.device("cuda", 0)
        buf = torch.zeros(len(full) + 4, dtype=torch.int64, device=dev); n_out = torch.zeros(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        e.scan_range_dev(f, -100 << sh, 100 << sh, buf, len(full), n_out); e.sync()
        assert int(n_out.item()) == len(full) and np.array_equal(buf[:len(full)].cpu().numpy().view(np.uint64), full) and int(buf[len(full):].abs().sum().item()) == 0
        e.index_set_ordered(f, 0)       # off: the column scan answers again, in position order
        assert e.index_ordered_info(f)[:2] == (0, False)
        pos = e.scan_range_pos(f, -50 << sh, 50 << sh)
        assert np.all(np.diff(pos.astype(np.int64)) > 0)
        _check(e, f, ids, vals, -50 << sh, 50 << sh)


def test_the_view_follows_every_change_of_its_field_and_nothing_else():
    R = 400_000
    rng = np.random.default_rng(11)
    ids = streams.splitmix64_np(np.arange(1, R + 1, dtype=np.uint64))
    vals = rng.integers(0, 300, R).astype(np.int64)
    ts = np.full(R, 10, np.int64)
    alive = np.ones(R, bool)
    with bmx.Engine(4 * R) as e:
        e.load_rows(ids, np.full(R, FA, np.uint32), ts, vals)
        e.load_rows(ids, np.full(R, FO, np.uint32), ts, vals + 7)
        e.index_build(FA)
        e.index_set_ordered(FA, 2)                      # a stale view is sorted again by the SECOND query after a change
        _q(e, FA, ids, vals, 10, 20); assert e.index_ordered_info(FA) == (2, False, 0)       # first query: scanned
        _q(e, FA, ids, vals, 10, 20); assert e.index_ordered_info(FA) == (2, True, 1)        # second: sorted
        _check(e, FA, ids, vals, 0, 299); assert e.index_ordered_info(FA) == (2, True, 1)
        # a merge on ANOTHER field: the view stays
        k = rng.choice(R, 50_000, replace=False)
        e.merge_batch(ids[k], np.full(len(k), FO, np.uint32), np.full(len(k), 20, np.int64), rng.integers(0, 300, len(k)).astype(np.int64))
        _check(e, FA, ids, vals, 100, 110); assert e.index_ordered_info(FA) == (2, True, 1)
        # a merge that rewrites the SAME values under a newer clock: rows win, no value changes, the view stays
        e.merge_batch(ids[k], np.full(len(k), FA, np.uint32), np.full(len(k), 20, np.int64), vals[k])
        _check(e, FA, ids, vals, 100, 110); assert e.index_ordered_info(FA) == (2, True, 1)
        # values change, rows are created, rows are tombstoned
        for rnd in range(3):
            k = rng.choice(R, 30_000, replace=False)
            nv = rng.integers(0, 300, len(k)).astype(np.int64)
            e.merge_batch(ids[k], np.full(len(k), FA, np.uint32), np.full(len(k), 1000 * (rnd + 1), np.int64), nv)     # (above every clock stored so far, tombstones included)
            vals[k] = nv; alive[k] = True
            new_ids = streams.splitmix64_np(np.arange(10_000_000 + rnd * 1000, 10_000_000 + rnd * 1000 + 500, dtype=np.uint64))
            new_vals = rng.integers(0, 300, 500).astype(np.int64)
            e.merge_batch(new_ids, np.full(500, FA, np.uint32), np.full(500, 5, np.int64), new_vals)
            ids = np.concatenate([ids, new_ids]); vals = np.concatenate([vals, new_vals]); alive = np.concatenate([alive, np.ones(500, bool)])
            d = rng.choice(R, 2_000, replace=False)
            e.put_rows(ids[d], np.full(len(d), FA, np.uint32), np.full(len(d), 1000 * (rnd + 1) + 1, np.int64), np.full(len(d), VAL_DELETED, np.int64))
            alive[d] = False
            s0 = e.index_ordered_info(FA)[2]
            p0 = e.index_ordered_stats(FA)["patches"]
            _q(e, FA, ids, vals, 50, 60, alive); assert e.index_ordered_info(FA) == (2, True, s0)        # round 5: the refresh patched the view, nothing was sorted
            assert e.index_ordered_stats(FA)["patches"] == p0 + 1
            for lo, hi in [(0, 0), (0, 299), (299, 299), (120, 180), (-5, 3)]:
                _check(e, FA, ids, vals, lo, hi, alive)
            assert e.index_ordered_info(FA) == (2, True, s0) and e.index_ordered_stats(FA)["patches"] == p0 + 1
        # a value that does not fit int32 arrives: the index switches to its 8-byte column, and the view is sorted from THAT
        k = rng.choice(R, 3, replace=False)
        wide = np.array([1 << 40, -(1 << 41), (1 << 31)], np.int64)
        e.merge_batch(ids[k], np.full(3, FA, np.uint32), np.full(3, 50_000, np.int64), wide)
        vals[k] = wide; alive[k] = True
        _q(e, FA, ids, vals, 0, 299, alive); _q(e, FA, ids, vals, 0, 299, alive)
        assert e.index_ordered_info(FA)[1]
        for lo, hi in [(1 << 40, 1 << 40), (-(1 << 42), -1), (1 << 31, 1 << 31), (250, 1 << 41), (0, 10)]:
            _check(e, FA, ids, vals, lo, hi, alive)
        # a table growth rebuilds the index (positions renumbered): the view goes with it
        e.reserve(16 * R)
        s0 = e.index_ordered_info(FA)[2]
        _q(e, FA, ids, vals, 7, 9, alive); _q(e, FA, ids, vals, 7, 9, alive)
        assert e.index_ordered_info(FA) == (2, True, s0 + 1)
        _check(e, FA, ids, vals, 7, 9, alive)
        e.index_drop(FA)
        _check(e, FA, ids, vals, 7, 9, alive)           # a query re-creates the index, without a view


def _merge(e, f, k_ids, ts, nv):
    e.merge_batch(k_ids, np.full(len(k_ids), f, np.uint32), np.full(len(k_ids), ts, np.int64), nv)


@pytest.mark.parametrize("wide", [False, True])
def test_a_current_view_stays_current_under_interleaved_merges_and_queries(wide):
    """VERDICT r4 item 4 (src/bullet-query.js:139-176: the reference moves a path between value buckets on every write): merges, puts, creations and
    tombstones on the indexed field interleaved with queries — every query is answered from the view (ids, positions, counts equal numpy over the
    model), NOTHING is ever sorted again, every refresh that changed something patched the view once. Rows changed twice between two queries, rows
    changed back to the value they had, duplicate keys inside a batch, revived tombstones, appended rows that sort in front of / behind everything."""
    R = 2_400_000                    # ord_n / 16 = 150k keys: a round's change run (~90k keys) first joins the view's PENDING patch, the next one makes main be rewritten
    sh = 34 if wide else 0
    rng = np.random.default_rng(21 + wide)
    ids = streams.splitmix64_np(np.arange(1, R + 1, dtype=np.uint64))
    vals = rng.integers(0, 500, R).astype(np.int64) << sh
    alive = np.ones(R, bool)
    F = FS if wide else FA
    clock = 10
    with bmx.Engine(4 * R) as e:
        e.load_rows(ids, np.full(R, F, np.uint32), np.full(R, clock, np.int64), vals)
        e.index_build(F); e.index_set_ordered(F, 1)
        _q(e, F, ids, vals, 3 << sh, 9 << sh)
        assert e.index_ordered_info(F) == (1, True, 1)
        patches = 0
        seen_pending = seen_rewrite = 0
        for rnd in range(8):
            n_now = len(ids)
            snap = vals.copy()
            # (a) a merge that changes values (duplicate keys inside the batch: the last-writer rule picks one)
            k = rng.choice(n_now, 40_000, replace=True)
            nv = rng.integers(0, 500, len(k)).astype(np.int64) << sh
            clock += 10
            tsb = np.full(len(k), clock, np.int64)
            e.merge_batch(ids[k], np.full(len(k), F, np.uint32), tsb, nv)
            best = {}
            for j in range(len(k)):          # equal clocks inside the batch: the larger value wins (src/bullet-crt.js:200-233)
                if int(k[j]) not in best or int(nv[j]) > best[int(k[j])]:
                    best[int(k[j])] = int(nv[j])
            kk = np.fromiter(best.keys(), np.int64); vv = np.fromiter(best.values(), np.int64)
            vals[kk] = vv; alive[kk] = True
            if rnd % 2 == 1:
                # (b) a SECOND merge before any query: some of the same rows change again, some go back to the value the view still holds
                k2 = kk[: len(kk) // 3]
                clock += 10
                nv2 = rng.integers(0, 500, len(k2)).astype(np.int64) << sh
                nv2[::2] = snap[k2[::2]]                         # ... back to what the view holds: no key moves for these rows after all
                _merge(e, F, ids[k2], clock, nv2); vals[k2] = nv2
            # (c) new rows: below every value, above every value, in the middle
            newn = 700
            new_ids = streams.splitmix64_np(np.arange(50_000_000 + rnd * 10_000, 50_000_000 + rnd * 10_000 + newn, dtype=np.uint64))
            new_vals = np.concatenate([np.full(100, -5 - rnd, np.int64) << sh, np.full(100, 900 + rnd, np.int64) << sh, rng.integers(0, 500, newn - 200).astype(np.int64) << sh])
            clock += 10
            _merge(e, F, new_ids, 5, new_vals)
            ids = np.concatenate([ids, new_ids]); vals = np.concatenate([vals, new_vals]); alive = np.concatenate([alive, np.ones(newn, bool)])
            # (d) tombstones, some on rows that (a) just changed
            d = np.concatenate([rng.choice(n_now, 1_500, replace=False), kk[:200]])
            d = np.unique(d)
            clock += 10
            e.put_rows(ids[d], np.full(len(d), F, np.uint32), np.full(len(d), clock, np.int64), np.full(len(d), VAL_DELETED, np.int64))
            alive[d] = False
            # the first query after the writes: patched, current, nothing sorted
            _q(e, F, ids, vals, 40 << sh, 60 << sh, alive)
            patches += 1
            st = e.index_ordered_stats(F)
            assert e.index_ordered_info(F) == (1, True, 1) and st["sorts"] == 1 and st["patches"] == patches, (rnd, st)
            seen_pending += st["pending_keys"] > 0; seen_rewrite = st["rewrites"]
            top = int(vals.max())
            for lo, hi in [(0, 0), (0, 499 << sh), (-(1 << 50), 1 << 50), ((-5 - rnd) << sh, (-5 - rnd) << sh), (top, top), (250 << sh, 251 << sh), (7 << sh, 6 << sh)]:
                _check(e, F, ids, vals, lo, hi, alive)
            assert e.index_ordered_stats(F)["patches"] == patches           # queries do not patch
            # a merge on this field that LOSES everywhere (old clock): the view is left alone
            k3 = rng.choice(len(ids), 5_000, replace=False)
            _merge(e, F, ids[k3], 1, (vals[k3] + (1 << sh)))
            _check(e, F, ids, vals, 100 << sh, 130 << sh, alive)
            assert e.index_ordered_stats(F)["patches"] == patches
        assert seen_pending >= 2 and seen_rewrite >= 2, (seen_pending, seen_rewrite)         # both states of the view were queried: a pending patch beside main, and main rewritten
        # the view equals what a fresh sort of the same columns gives: drop it, sort again, compare a whole-range position listing (element for element when the
        # patched view is one run; as a set while it answers from main + a pending patch: survivors of main first, then the inserted keys)
        one_run = e.index_ordered_stats(F)["pending_keys"] == 0
        pos_patched = e.scan_range_pos(F, -(1 << 50), 1 << 50)
        e.index_set_ordered(F, 0); e.index_set_ordered(F, 1)
        pos_sorted = e.scan_range_pos(F, -(1 << 50), 1 << 50)
        assert e.index_ordered_stats(F)["sorts"] == 2 and e.index_ordered_info(F)[1]       # (a new view: sorted from the columns as they are now)
        assert np.array_equal(pos_patched, pos_sorted) if one_run else np.array_equal(np.sort(pos_patched), np.sort(pos_sorted))


def test_every_patch_rewrites_main_when_the_pending_patch_is_switched_off(monkeypatch):
    """BMX_VIEW_PENDING=0 (A/B switch, read at create): every refresh merges its change run into the view's main run at once; after every round the patched view
    lists its positions in exactly the order a fresh sort gives (the streaming merge kernel, element for element)"""
    monkeypatch.setenv("BMX_VIEW_PENDING", "0")
    R = 1_500_000
    rng = np.random.default_rng(77)
    ids = streams.splitmix64_np(np.arange(1, R + 1, dtype=np.uint64))
    vals = rng.integers(0, 2000, R).astype(np.int64)
    with bmx.Engine(3 * R) as e:
        monkeypatch.delenv("BMX_VIEW_PENDING")
        e.load_rows(ids, np.full(R, FA, np.uint32), np.full(R, 5, np.int64), vals)
        e.index_build(FA)
        e.index_set_ordered(FA, 1)
        assert e.scan_count(FA, 0, 10) == int((vals <= 10).sum())
        for rnd in range(4):
            k = rng.choice(len(ids), 20_000 * (rnd + 1), replace=False)
            nv = rng.integers(0, 2000, len(k)).astype(np.int64)
            new_ids = streams.splitmix64_np(np.arange(70_000_000 + rnd * 5000, 70_000_000 + rnd * 5000 + 3000, dtype=np.uint64))
            new_vals = rng.integers(0, 2000, 3000).astype(np.int64)
            _merge(e, FA, ids[k], 100 + rnd, nv); _merge(e, FA, new_ids, 5, new_vals)
            vals[k] = nv; ids = np.concatenate([ids, new_ids]); vals = np.concatenate([vals, new_vals])
            got = e.scan_range_pos(FA, -(1 << 40), 1 << 40).astype(np.int64)
            st = e.index_ordered_stats(FA)
            assert st["sorts"] == 1 and st["patches"] == rnd + 1 and st["rewrites"] == rnd + 1 and st["pending_keys"] == 0, st
            # the whole view, in order: every row once, keys (value, position) strictly ascending — what a fresh sort of the columns gives, element for element
            col = e.index_ids(FA)
            order = np.argsort(ids); v_of = vals[order][np.searchsorted(ids[order], col[got])]
            assert len(got) == len(ids) and np.array_equal(np.sort(col[got]), ids[order])
            dv, dp = np.diff(v_of), np.diff(got)
            assert np.all((dv > 0) | ((dv == 0) & (dp > 0))), rnd


def test_patch_of_a_large_view_with_a_skewed_change_run():
    """20M rows, a 1M-delta batch whose new values all land in ONE narrow stretch of the value domain (every inserted key falls into a few tiles of the
    view: the LDS window of the streaming merge overflows there and the global-memory search takes over), plus rows appended behind everything"""
    R = 20_000_000
    rng = np.random.default_rng(33)
    ids = streams.splitmix64_np(np.arange(1, R + 1, dtype=np.uint64))
    vals = rng.integers(0, 1000, R).astype(np.int64)
    with bmx.Engine(R + 4_000_000) as e:
        e.load_rows(ids, np.full(R, FA, np.uint32), np.full(R, 5, np.int64), vals)
        e.index_build(FA); e.index_set_ordered(FA, 1)
        assert e.scan_count(FA, 10, 19) == int(((vals >= 10) & (vals <= 19)).sum())
        k = rng.choice(R, 1_000_000, replace=False)
        nv = np.full(len(k), 777, np.int64); nv[::3] = 778
        _merge(e, FA, ids[k], 50, nv); vals[k] = nv
        new_ids = streams.splitmix64_np(np.arange(900_000_000, 900_000_000 + 300_000, dtype=np.uint64))
        new_vals = np.full(300_000, 5000, np.int64)
        _merge(e, FA, new_ids, 5, new_vals)
        ids = np.concatenate([ids, new_ids]); vals = np.concatenate([vals, new_vals])
        for lo, hi in [(777, 777), (778, 778), (0, 776), (779, 999), (5000, 5000), (0, 1 << 40)]:
            assert e.scan_count(FA, lo, hi) == int(((vals >= lo) & (vals <= hi)).sum()), (lo, hi)
        st = e.index_ordered_stats(FA)
        assert st["sorts"] == 1 and st["patches"] == 1, st
        got = e.scan_range(FA, 777, 778)
        assert np.array_equal(np.sort(got), np.sort(ids[(vals >= 777) & (vals <= 778)]))
        pos = e.scan_range_pos(FA, 5000, 5000)
        assert np.array_equal(np.sort(e.index_ids(FA)[pos]), np.sort(new_ids))


def test_without_patching_a_field_written_between_any_two_queries_never_pays_for_a_sort(monkeypatch):
    """ADVICE r4 (bmx.hip stale_queries): with N = 2 and write / query / write / query ... the stale-query count belongs to ONE change of the columns
    and starts again with the next; it used to count since the last SORT, so every second query sorted a view the next write threw away. Patching is
    switched off for this context (BMX_VIEW_PATCH=0, read at create): this is the policy for views that cannot be patched."""
    monkeypatch.setenv("BMX_VIEW_PATCH", "0")
    R = 300_000
    rng = np.random.default_rng(13)
    ids = streams.splitmix64_np(np.arange(1, R + 1, dtype=np.uint64))
    vals = rng.integers(0, 300, R).astype(np.int64)
    with bmx.Engine(4 * R) as e:
        monkeypatch.delenv("BMX_VIEW_PATCH")
        e.load_rows(ids, np.full(R, FA, np.uint32), np.full(R, 5, np.int64), vals)
        e.index_build(FA); e.index_set_ordered(FA, 2)
        _q(e, FA, ids, vals, 1, 2); _q(e, FA, ids, vals, 1, 2)
        assert e.index_ordered_info(FA) == (2, True, 1)
        for rnd in range(6):
            k = rng.choice(R, 10_000, replace=False)
            nv = rng.integers(0, 300, len(k)).astype(np.int64)
            _merge(e, FA, ids[k], 100 + rnd, nv); vals[k] = nv
            _q(e, FA, ids, vals, 10, 40)
            assert e.index_ordered_info(FA) == (2, False, 1), rnd           # one query per change: scanned every time, never sorted
        assert e.index_ordered_stats(FA)["patches"] == 0
        _q(e, FA, ids, vals, 10, 40)                                        # the SECOND query since the last change sorts
        assert e.index_ordered_info(FA) == (2, True, 2)


def test_small_and_degenerate_columns():
    with bmx.Engine(10_000) as e:
        ids = streams.splitmix64_np(np.arange(1, 201, dtype=np.uint64))
        vals = np.full(200, 42, np.int64)                  # one value, 200 times
        e.load_rows(ids, np.full(200, FA, np.uint32), np.full(200, 5, np.int64), vals)
        e.index_build(FA); e.index_set_ordered(FA, 1)
        for lo, hi in [(42, 42), (41, 43), (43, 50), (0, 41), (-1 << 40, 1 << 40)]:
            _check(e, FA, ids, vals, lo, hi)
        one = streams.splitmix64_np(np.arange(900, 901, dtype=np.uint64))
        e.load_rows(one, np.full(1, FS, np.uint32), np.full(1, 5, np.int64), np.array([-(1 << 40)], np.int64))      # a single wide row
        e.index_build(FS); e.index_set_ordered(FS, 1)
        _check(e, FS, one, np.array([-(1 << 40)], np.int64), -(1 << 41), 0)
        _check(e, FS, one, np.array([-(1 << 40)], np.int64), 0, 5)
        assert e.index_ordered_info(FS)[1]


def test_auto_policy_sorts_once_the_scans_have_cost_a_sort():
    """BMX_INDEX_ORDERED_AUTO: rent or buy — a 400k-row int32 column scans in ~8.3 us, its first sort is priced at 224 us: the 28th query since the change sorts"""
    R = 400_000
    rng = np.random.default_rng(5)
    ids = streams.splitmix64_np(np.arange(1, R + 1, dtype=np.uint64))
    vals = rng.integers(0, 1000, R).astype(np.int64)
    with bmx.Engine(2 * R) as e:
        e.load_rows(ids, np.full(R, FA, np.uint32), np.full(R, 5, np.int64), vals)
        e.index_build(FA)
        e.index_set_ordered(FA, 0xFFFFFFFF)
        for i in range(27):
            _q(e, FA, ids, vals, i, i + 10)
        assert e.index_ordered_info(FA) == (0xFFFFFFFF, False, 0)
        _q(e, FA, ids, vals, 5, 50)
        assert e.index_ordered_info(FA) == (0xFFFFFFFF, True, 1)
        _check(e, FA, ids, vals, 100, 300)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 66, 4095, 4096, 4097, 4161, 262143, 262144, 262145, 266305])
def test_search_rounds_at_their_boundaries(n):
    """the 64-ary search narrows [L, R) by a factor of 64 per round and finishes with one wave over <= 64 keys: column sizes on both sides of every
    round count, values with long runs of duplicates, every bound from below the smallest to above the largest value"""
    rng = np.random.default_rng(n)
    ids = streams.splitmix64_np(np.arange(1, n + 1, dtype=np.uint64))
    vals = (rng.integers(0, max(2, n // 50), n) * 3).astype(np.int64)           # multiples of 3: keys between the values exist too
    with bmx.Engine(max(4096, 2 * n)) as e:
        e.load_rows(ids, np.full(n, FA, np.uint32), np.full(n, 5, np.int64), vals)
        e.index_build(FA); e.index_set_ordered(FA, 1)
        top = int(vals.max())
        keys = sorted(set([-4, -1, 0, 1, 2, 3, top - 1, top, top + 1, top + 7] + rng.integers(-3, top + 4, 24).tolist()))
        for lo in keys[::3]:
            for hi in keys[::2]:
                want = int(((vals >= lo) & (vals <= hi)).sum())
                assert e.scan_count(FA, lo, hi) == want, (n, lo, hi)
        for lo, hi in [(0, top), (3, 3), (top, top), (1, 2), (top + 1, top + 9)]:
            _q(e, FA, ids, vals, lo, hi)
        assert e.index_ordered_info(FA)[1:] == (True, 1)


def test_declarative_filter_takes_its_candidates_from_the_view():
    """bmx_scan_filter (AND of range terms over fields of one node, subset of src/bullet-query.js:270-283): with a view on the first term's index the
    other terms are probed for the ids of ONE run only; same set as the column form, count-only and truncated forms included"""
    R = 300_000
    rng = np.random.default_rng(9)
    ids = streams.splitmix64_np(np.arange(1, R + 1, dtype=np.uint64))
    age = rng.integers(0, 100, R).astype(np.int64); score = rng.integers(-500, 500, R).astype(np.int64); other = rng.integers(0, 10, R).astype(np.int64)
    with bmx.Engine(4 * R) as e:
        for f, v in ((FA, age), (FS, score), (FO, other)):
            e.load_rows(ids, np.full(R, f, np.uint32), np.full(R, 5, np.int64), v)
        some = rng.choice(R, 5000, replace=False)                      # nodes without a score row at all: they can match no term on it
        e.put_rows(ids[some], np.full(len(some), FS, np.uint32), np.full(len(some), 9, np.int64), np.full(len(some), VAL_DELETED, np.int64))
        has_score = np.ones(R, bool); has_score[some] = False
        e.index_build(FA)
        qs = [[(FA, 30, 40), (FS, 0, 100)], [(FA, 0, 99), (FS, -10, 10), (FO, 3, 5)], [(FA, 50, 50)], [(FA, 98, 200), (FO, 0, 0)], [(FA, 5, 4), (FS, 0, 1)]]

        def want(q):
            m = np.ones(R, bool)
            for f, lo, hi in q:
                v = {FA: age, FS: score, FO: other}[f]
                m &= (v >= lo) & (v <= hi)
                if f == FS:
                    m &= has_score
            return np.sort(ids[m])
        plain = [np.sort(e.scan_filter(q)) for q in qs]
        e.index_set_ordered(FA, 1)
        for q, p0 in zip(qs, plain):
            got = e.scan_filter(q)
            assert np.array_equal(np.sort(got), want(q)) and np.array_equal(np.sort(got), p0), q
            few = e.scan_filter(q, cap=7)
            assert len(few) == min(7, len(p0)) and np.all(np.isin(few, p0))
        assert e.index_ordered_info(FA)[1:] == (True, 1)


def test_two_views_of_one_context_are_patched_by_the_same_refresh_and_filters_see_the_patch():
    """Two ordered indexes of ONE context — an int32 column and a wide int64 column — written by the same merges: one refresh of the change log patches both (they share the
    sort scratch, the result words and the one background rewrite the context allows at a time). After every round both answer ranges, counts, positions and the declarative
    filter (candidates from the first term's view, pending patch included) like numpy; a large host-mode answer (beyond the small-answer path) comes back whole."""
    R = 1_200_000                       # ord_n / 16 = 75k keys: rounds of ~50k changed rows per field go through the pending patch and its rewrite
    rng = np.random.default_rng(515)
    ids = streams.splitmix64_np(np.arange(1, R + 1, dtype=np.uint64))
    age = rng.integers(0, 300, R).astype(np.int64)
    score = rng.integers(-1000, 1000, R).astype(np.int64) << 33
    with bmx.Engine(4 * R) as e:
        e.load_rows(ids, np.full(R, FA, np.uint32), np.full(R, 5, np.int64), age)
        e.load_rows(ids, np.full(R, FS, np.uint32), np.full(R, 5, np.int64), score)
        e.index_build(FA); e.index_build(FS)
        e.index_set_ordered(FA, 1); e.index_set_ordered(FS, 1)
        _q(e, FA, ids, age, 10, 20); _q(e, FS, ids, score, 0, 5 << 33)
        states = set()
        for rnd in range(7):
            k = rng.choice(R, 25_000, replace=False)
            na = rng.integers(0, 300, len(k)).astype(np.int64); ns = rng.integers(-1000, 1000, len(k)).astype(np.int64) << 33
            # ONE batch carries both fields' deltas (interleaved): both indexes see their winners in the same change log
            bi = np.concatenate([ids[k], ids[k]]); bf = np.concatenate([np.full(len(k), FA, np.uint32), np.full(len(k), FS, np.uint32)])
            bv = np.concatenate([na, ns]); perm = rng.permutation(len(bi))
            e.merge_batch(bi[perm], bf[perm], np.full(len(bi), 10 + 2 * rnd, np.int64), bv[perm])          # (clocks ascend over all merges of the test: every delta wins)
            age[k] = na; score[k] = ns
            if rnd % 3 == 2:                # a round that writes only ONE of the two fields: the other view must be left alone (no patch counted)
                _q(e, FA, ids, age, 1, 2)       # (this refresh patches BOTH views with the batch above)
                before = e.index_ordered_stats(FS)["patches"]
                k1 = rng.choice(R, 9_000, replace=False); n1 = rng.integers(0, 300, len(k1)).astype(np.int64)
                _merge(e, FA, ids[k1], 11 + 2 * rnd, n1); age[k1] = n1
                _q(e, FA, ids, age, 100, 110)
                assert e.index_ordered_stats(FS)["patches"] == before
            _check(e, FA, ids, age, 40, 60); _check(e, FS, ids, score, -(20 << 33), 20 << 33)
            sa, ss = e.index_ordered_stats(FA), e.index_ordered_stats(FS)
            assert sa["sorts"] == 1 and ss["sorts"] == 1 and e.index_ordered_info(FA)[1] and e.index_ordered_info(FS)[1], (rnd, sa, ss)
            states.add((sa["pending_keys"] > 0, ss["pending_keys"] > 0))
            big = e.scan_range(FA, 0, 299)                                   # every row: a host-mode answer far beyond the small-answer buffer
            assert len(big) == R and np.array_equal(np.sort(big), np.sort(ids))
            for q in ([(FA, 40, 45), (FS, -(100 << 33), 100 << 33)], [(FS, 0, 3 << 33), (FA, 0, 150)], [(FA, 299, 299)]):
                m = np.ones(R, bool)
                for f, lo, hi in q:
                    v = age if f == FA else score
                    m &= (v >= lo) & (v <= hi)
                got = e.scan_filter(q)
                assert np.array_equal(np.sort(got), np.sort(ids[m])), (rnd, q, len(got), int(m.sum()))
        assert (True, True) in states and len(states) >= 2, states         # both views were queried with a patch pending, and in at least one other state
        assert e.index_ordered_stats(FA)["rewrites"] >= 1 and e.index_ordered_stats(FS)["rewrites"] >= 1


@pytest.mark.parametrize("wide", [False, True])
def test_a_view_sorted_by_the_patch_paths_own_kernels_equals_the_library_sorted_one(monkeypatch, wide):
    """BMX_VIEW_SORT=own (A/B switch, read at create): a new view is sorted by k_view_tile_sort + k_view_merge_pass over the whole column instead of rocPRIM's radix
    sort. Both arms list the positions in exactly (value, position) order of their own columns, element for element — tombstones in front, a ragged last tile, more rows than one pass covers."""
    sh = 35 if wide else 0
    R = 1_300_077
    rng = np.random.default_rng(8 + wide)
    ids = streams.splitmix64_np(np.arange(1, R + 1, dtype=np.uint64))
    vals = (rng.integers(-700, 700, R).astype(np.int64)) << sh
    dead = rng.choice(R, 3000, replace=False)
    F = FS if wide else FA
    listings = []
    for arm in ("own", "library"):
        if arm == "own":
            monkeypatch.setenv("BMX_VIEW_SORT", "own")
        else:
            monkeypatch.delenv("BMX_VIEW_SORT", raising=False)
        with bmx.Engine(2 * R) as e:
            e.load_rows(ids, np.full(R, F, np.uint32), np.full(R, 5, np.int64), vals)
            e.put_rows(ids[dead], np.full(len(dead), F, np.uint32), np.full(len(dead), 9, np.int64), np.full(len(dead), VAL_DELETED, np.int64))
            e.index_build(F); e.index_set_ordered(F, 1)
            alive = np.ones(R, bool); alive[dead] = False
            _check(e, F, ids, vals, -(3 << sh), 4 << sh, alive)
            assert e.index_ordered_info(F)[1:] == (True, 1)
            got = e.scan_range_pos(F, -(1 << 60), 1 << 60).astype(np.int64)
            col = e.index_ids(F)
            order = np.argsort(ids); v_at = vals[order][np.searchsorted(ids[order], col)]        # value of the row at every index position
            live = np.flatnonzero(np.isin(col, ids[alive]))
            want = live[np.lexsort((live, v_at[live]))]                                           # (value, position) ascending
            bad = np.flatnonzero(got != want) if len(got) == len(want) else np.array([-1])
            assert len(bad) == 0, (arm, len(got), len(want), bad[:5], got[bad[:5]] if bad[0] >= 0 else None, want[bad[:5]] if bad[0] >= 0 else None)
            listings.append(len(got))           # (two engines may number their index positions differently: each arm is held against its own columns)
    assert listings[0] == listings[1] == R - len(dead)


@pytest.mark.parametrize("mode", ["1", "2"])
def test_failures_of_the_patch_and_of_the_background_rewrite_cost_time_never_answers(monkeypatch, mode):
    """BMX_TEST_VIEW_FAIL (test hook, read at create). 1: every patch reports failure — the view goes stale, the queries scan the column, the N-th re-sorts, exactly round 4's
    behaviour. 2: every background rewrite of main reports failure — main and the pending patch were never written and go on answering, the patch keeps growing past its
    threshold (its buffers grow with it), every later patch tries again. Answers equal numpy throughout."""
    monkeypatch.setenv("BMX_TEST_VIEW_FAIL", mode)
    R = 1_000_000                       # ord_n / 16 = 62.5k keys
    rng = np.random.default_rng(40 + int(mode))
    ids = streams.splitmix64_np(np.arange(1, R + 1, dtype=np.uint64))
    vals = rng.integers(0, 400, R).astype(np.int64)
    with bmx.Engine(3 * R) as e:
        monkeypatch.delenv("BMX_TEST_VIEW_FAIL")
        e.load_rows(ids, np.full(R, FA, np.uint32), np.full(R, 5, np.int64), vals)
        e.index_build(FA); e.index_set_ordered(FA, 2)
        _q(e, FA, ids, vals, 5, 9); _q(e, FA, ids, vals, 5, 9)                 # the second query sorts (N = 2)
        assert e.index_ordered_info(FA) == (2, True, 1)
        pend = []
        for rnd in range(7):
            k = rng.choice(R, 30_000, replace=False)
            nv = rng.integers(0, 400, len(k)).astype(np.int64)
            _merge(e, FA, ids[k], 10 + rnd, nv); vals[k] = nv
            _check(e, FA, ids, vals, 100, 140)
            st = e.index_ordered_stats(FA)
            if mode == "1":
                assert st["patches"] == 0 and not e.index_ordered_info(FA)[1] or st["sorts"] >= 2, (rnd, st)     # stale until the second query since the change sorts again
                _q(e, FA, ids, vals, 0, 399)
            else:
                assert st["sorts"] == 1 and st["patches"] == rnd + 1 and st["rewrites"] == 0 and e.index_ordered_info(FA)[1], (rnd, st)
                pend.append(st["pending_keys"])
            _check(e, FA, ids, vals, 0, 0); _check(e, FA, ids, vals, 399, 399)
        if mode == "1":
            assert e.index_ordered_stats(FA)["sorts"] >= 4
        else:
            assert pend == sorted(pend) and pend[-1] > 3 * (R // 16), pend          # nothing was ever folded into main: the patch only grows, well past its threshold


def test_host_mode_answers_written_straight_into_a_page_locked_caller_buffer():
    """A host-mode scan whose out buffer lives in page-locked memory (bmx_host_alloc) is written by the kernels themselves, no staging copy: column scan and view, answers
    beyond the small-answer path, a cap smaller than the answer, and the same calls into ordinary (pageable) arrays."""
    R = 600_000
    rng = np.random.default_rng(3)
    ids = streams.splitmix64_np(np.arange(1, R + 1, dtype=np.uint64))
    vals = rng.integers(0, 50, R).astype(np.int64)
    hb = bmx.HostBuffer(R * 8)
    with bmx.Engine(2 * R) as e:
        e.load_rows(ids, np.full(R, FA, np.uint32), np.full(R, 5, np.int64), vals)
        e.index_build(FA)
        for ordered in (0, 1):
            e.index_set_ordered(FA, ordered)
            for lo, hi in [(0, 49), (10, 19), (7, 7), (60, 70)]:
                want = np.sort(ids[(vals >= lo) & (vals <= hi)])
                pinned = hb.array(np.uint64, R); pinned[:] = 0
                got = e.scan_range(FA, lo, hi, out=pinned)
                assert len(got) == len(want) and np.array_equal(np.sort(got), want), (ordered, lo, hi)
                plain = np.zeros(R, np.uint64)
                got2 = e.scan_range(FA, lo, hi, out=plain)
                assert np.array_equal(np.sort(got2), want)
            small = hb.array(np.uint64, 20_000); small[:] = 0                     # room for fewer ids than match: the first 20000 of the answer, the count says how many there are
            got = e.scan_range(FA, 0, 49, out=small)
            assert len(got) == 20_000 and np.all(np.isin(got, ids)) and len(np.unique(got)) == 20_000
            assert e.scan_count(FA, 0, 49) == R
    hb.close()
