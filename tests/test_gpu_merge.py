"""GPU parity: HIP merge path (through the C ABI) vs the CPU oracle and the reference golden vectors.
Bit-exact: integer timestamps/values, deterministic tie-breaks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import bmx
from oracle import streams
from bmx import synth
from oracle.oracle import Oracle, INSERT_REFERENCE, INSERT_DELTA, rows_digest
from helpers import load_golden, stream_fixtures, golden_flags, run_oracle_stream

F0 = streams.field_hash(0)


def _state(e):
    id, f, ts, val = e.dump_rows()
    o = np.lexsort((f, id))
    return id[o], f[o], ts[o], val[o]


def _ostate(o):
    id, f, ts, val = o.dump_rows()
    k = np.lexsort((f, id))
    return id[k], f[k], ts[k], val[k]


def _assert_same_state(e, o):
    a, b = _state(e), _ostate(o)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_library_reports_abi_and_device():
    with bmx.Engine(1000) as e:
        i = e.info()
        assert i.abi_version == 4 and i.n_rows == 0 and i.n_slots >= 2000 and i.table_bytes == i.n_slots * 32


def test_decision_table_golden():
    """Every row of resolve()'s decision table (src/bullet-crt.js:164-279), one key per case in one batch."""
    g = load_golden("g1_decision_table.json")
    cases = g["cases"]
    n = len(cases)
    ids = streams.splitmix64_np(np.arange(1, n + 1, dtype=np.uint64))
    with bmx.Engine(4 * n) as e:
        res = [i for i, c in enumerate(cases) if c["cur"] is not None]
        e.load_rows(ids[res], np.full(len(res), F0), [cases[i]["cur"][0] for i in res], [cases[i]["cur"][1] for i in res])
        applied, flags, st = e.merge_batch(ids, np.full(n, F0), [c["inc"][0] for c in cases], [c["inc"][1] for c in cases])
        assert st.n_conflicts == 0
        assert flags.tolist() == [c["flags"] for c in cases]
        assert applied.tolist() == [i for i, c in enumerate(cases) if c["flags"] & 1]
        ts, val, found = e.get_rows(ids, np.full(n, F0))
        assert found.all()
        assert ts.tolist() == [c["out"][0] for c in cases]
        assert val.tolist() == [c["out"][1] for c in cases]


def test_sequences_golden_duplicates_and_insert_quirk():
    """300 one-key sequences (ts 0..4, val -1..1) as ONE batch: every key has duplicates inside the batch."""
    g = load_golden("g3_sequences.json")
    ids, ts, val, owner = [], [], [], []
    res_ids, res_ts, res_val = [], [], []
    for k, s in enumerate(g["seqs"]):
        kid = streams.splitmix64(1000 + k)
        if s["start"] is not None:
            res_ids.append(kid); res_ts.append(s["start"][0]); res_val.append(s["start"][1])
        for d in s["deltas"]:
            ids.append(kid); ts.append(d[0]); val.append(d[1]); owner.append(k)
    # interleave the keys round-robin so duplicates of one key sit in different waves and blocks,
    # keeping each key's own order
    n = len(ids)
    per_key_src = {}
    for j in range(n):
        per_key_src.setdefault(owner[j], []).append(j)
    out_ids = np.zeros(n, np.uint64); out_ts = np.zeros(n, np.int64); out_val = np.zeros(n, np.int64)
    out_owner = np.zeros(n, np.int64); out_local = np.zeros(n, np.int64)
    cursors = {k: 0 for k in per_key_src}
    j = 0
    while j < n:
        for k in per_key_src:
            c = cursors[k]
            if c < len(per_key_src[k]):
                s_ = per_key_src[k][c]
                out_ids[j] = ids[s_]; out_ts[j] = ts[s_]; out_val[j] = val[s_]; out_owner[j] = k; out_local[j] = c
                cursors[k] = c + 1
                j += 1
    with bmx.Engine(8192) as e:
        e.load_rows(res_ids, np.full(len(res_ids), F0), res_ts, res_val)
        applied, flags, st = e.merge_batch(out_ids, np.full(n, F0), out_ts, out_val, INSERT_REFERENCE)
        assert st.n_conflicts > 0
        win = {int(out_owner[a]): int(out_local[a]) for a in applied.tolist()}
        assert len(win) == len(applied)
        for k, s in enumerate(g["seqs"]):
            kid = streams.splitmix64(1000 + k)
            assert list(e.get_row(kid, F0)) == s["final"], (k, s)
            assert win.get(k, -1) == s["winner"], (k, s)
        # and the oracle agrees on the interleaved batch as a whole
        o = Oracle(); o.load_rows(res_ids, np.full(len(res_ids), F0), res_ts, res_val)
        _, ow = o.merge_batch(out_ids, np.full(n, F0), out_ts, out_val, INSERT_REFERENCE)
        assert np.array_equal(applied, ow)
        _assert_same_state(e, o)


@pytest.mark.parametrize("name", stream_fixtures())
def test_stream_golden(name):
    g = load_golden(name)
    spec = g["spec"]
    st, o, oflags, owinners = run_oracle_stream(spec)
    _, rid, rf, rts, rval = st["resident"]
    _, did, df, dts, dval = st["deltas"]
    with bmx.Engine(max(1024, 2 * (spec["R"] + spec["D"]))) as e:
        e.load_rows(rid, rf, rts, rval)
        assert e.row_count() == spec["R"]
        applied, flags, stats = e.merge_batch(did, df, dts, dval, INSERT_REFERENCE)
        assert applied.tolist() == g["winners"]          # reference's final winner per changed key
        assert stats.n_applied == len(g["winners"])
        assert e.row_count() == g["n_rows_final"] == stats.n_rows
        id, f, ts, val = e.dump_rows()
        assert "%x" % rows_digest(id, f, ts, val) == g["digest"]   # reference's final state
        _assert_same_state(e, o)
        if stats.n_conflicts == 0:
            assert np.array_equal(flags, golden_flags(g))   # per-delta decision flags of the reference
        else:
            # flags of winners are always INCOMING
            assert (flags[applied] & 1).all()


def test_empty_batch_and_empty_table():
    with bmx.Engine(100) as e:
        applied, flags, st = e.merge_batch([], [], [], [])
        assert len(applied) == 0 and st.n_applied == 0
        assert e.row_count() == 0
        assert e.get_row(5, F0) is None
        id, f, ts, val = e.dump_rows()
        assert len(id) == 0


def test_domain_errors_are_reported_not_fatal():
    with bmx.Engine(100) as e:
        with pytest.raises(bmx.BmxError) as ei:
            e.merge_batch([2**64 - 1], [F0], [5], [1])       # reserved id
        assert ei.value.code == bmx.ERR_RANGE
        with pytest.raises(bmx.BmxError):
            e.merge_batch([7], [0xFFFFFFFF], [5], [1])       # reserved field
        with pytest.raises(bmx.BmxError):
            e.merge_batch([7], [F0], [-1], [1])              # negative clock
        with pytest.raises(bmx.BmxError):
            e.merge_batch([7], [F0], [2**53], [1])           # beyond JS safe integer
        with pytest.raises(bmx.BmxError):
            e.merge_batch([7], [F0], [5], [-(2**53)])
        # engine still usable; extreme legal values work
        applied, _, _ = e.merge_batch([7, 8], [F0, F0], [2**53 - 1, 0], [-(2**53 - 1), 2**53 - 1], INSERT_DELTA)
        assert applied.tolist() == [0, 1]
        assert e.get_row(7, F0) == (2**53 - 1, -(2**53 - 1))
        assert e.get_row(8, F0) == (0, 2**53 - 1)


def test_table_grows_by_rehash_and_keeps_every_row():
    """Default contexts grow (device-side rehash) instead of failing; scans and merges keep working across the growth."""
    o = Oracle()
    with bmx.Engine(1000) as e:
        n0 = e.info().n_slots
        for b in range(12):
            d = synth.big_deltas(3000, 1000, seed=77, insert_pct=70, hot_pct=10, hot_keys=20, unique=False, batch=b)
            applied, _, st = e.merge_batch(*d)
            _, ow = o.merge_batch(*d)
            assert np.array_equal(applied, ow), b
        assert e.row_count() == len(o) > 1000 and e.info().n_slots > n0
        _assert_same_state(e, o)
        f0 = int(d[1][0])
        assert np.array_equal(np.sort(e.scan_range(f0, -(1 << 40), 1 << 40)), np.sort(o.scan_range(f0, -(1 << 40), 1 << 40)))
        e.reserve(200_000)                       # explicit reservation: same rows afterwards
        assert e.info().capacity_rows == 200_000
        _assert_same_state(e, o)


def test_table_full_is_an_error_with_fixed_capacity():
    with bmx.Engine(1000, flags=bmx.CTX_FIXED_CAPACITY) as e:
        ids = streams.splitmix64_np(np.arange(1, 9001, dtype=np.uint64))
        with pytest.raises(bmx.BmxError) as ei:
            for k in range(0, 9000, 1000):
                e.merge_batch(ids[k:k + 1000], np.full(1000, F0), np.full(1000, 5), np.zeros(1000))
        assert ei.value.code == bmx.ERR_FULL


def test_delta_insert_mode_true_lww():
    o = Oracle()
    with bmx.Engine(1000) as e:
        ids = [1, 1, 1, 2, 2]; ts = [100, 50, 100, 7, 7]; val = [5, 7, 6, 1, 1]
        applied, flags, st = e.merge_batch(ids, [F0] * 5, ts, val, INSERT_DELTA)
        _, ow = o.merge_batch(ids, [F0] * 5, ts, val, INSERT_DELTA)
        assert applied.tolist() == ow.tolist() == [2, 3]
        assert e.get_row(1, F0) == (100, 6) and e.get_row(2, F0) == (7, 1)


def test_idempotent_and_commutative_properties_at_scale():
    """1M-row table, 200k-delta batches with hot keys: re-applying a batch changes nothing;
    applying two batches in either order gives the same state when no inserts are involved (true LWW on hits)."""
    R, D = 1_000_000, 200_000
    rid, rf, rts, rval = synth.big_resident(R, seed=3)
    b1 = synth.big_deltas(D, R, seed=4, insert_pct=0, hot_pct=30, hot_keys=R // 1000, unique=False, batch=0)
    b2 = synth.big_deltas(D, R, seed=5, insert_pct=0, hot_pct=30, hot_keys=R // 1000, unique=False, batch=1)
    with bmx.Engine(2 * R) as e1, bmx.Engine(2 * R) as e2:
        e1.load_rows(rid, rf, rts, rval); e2.load_rows(rid, rf, rts, rval)
        a1, _, s1 = e1.merge_batch(*b1); e1.merge_batch(*b2)
        e2.merge_batch(*b2); e2.merge_batch(*b1)
        assert s1.n_conflicts > 0
        d1 = rows_digest(*e1.dump_rows()); d2 = rows_digest(*e2.dump_rows())
        assert d1 == d2
        again, _, s = e1.merge_batch(*b1)
        assert len(again) == 0 and s.n_applied == 0
        assert rows_digest(*e1.dump_rows()) == d1
        o = Oracle(); o.load_rows(rid, rf, rts, rval)
        _, ow = o.merge_batch(*b1)
        assert np.array_equal(a1, ow)
        o.merge_batch(*b2)
        assert o.digest() == d1


def test_epoch_wrap_many_small_batches():
    """More than 255 batches: the 8-bit claim epoch wraps and heads are swept."""
    R = 2000
    rid, rf, rts, rval = synth.big_resident(R, seed=21)
    o = Oracle(); o.load_rows(rid, rf, rts, rval)
    with bmx.Engine(3 * R) as e:
        e.load_rows(rid, rf, rts, rval)
        for b in range(300):
            d = synth.big_deltas(64, R, seed=22, insert_pct=5, hot_pct=50, hot_keys=8, unique=False, batch=b, DT=40, T0=1_000_000)
            applied, _, _ = e.merge_batch(*d)
            _, ow = o.merge_batch(*d)
            assert np.array_equal(applied, ow), b
        _assert_same_state(e, o)
        assert e.info().epoch < 256


def test_all_deltas_one_key_worst_case_contention():
    n = 20000
    rng = np.random.default_rng(1)
    ts = rng.integers(0, 50, n); val = rng.integers(-3, 4, n)
    o = Oracle()
    _, ow = o.merge_batch(np.full(n, 42, np.uint64), np.full(n, F0), ts, val)
    with bmx.Engine(n) as e:
        applied, _, st = e.merge_batch(np.full(n, 42, np.uint64), np.full(n, F0), ts, val)
        assert np.array_equal(applied, ow) and len(applied) == 1
        assert e.get_row(42, F0) == o.get_row(42, F0)
        assert st.n_conflicts == n - 1


@pytest.mark.parametrize("mode", [INSERT_REFERENCE, INSERT_DELTA])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_duplicate_heavy_random_vs_oracle(seed, mode):
    """Lists of 1..60 same-key deltas (short and long resolve paths), ts in 0..4 and val in -1..1 so ties and the
    ts := 2 insert rule are hit constantly; half the keys resident, half absent; three consecutive batches."""
    rng = np.random.default_rng(seed)
    K = 3000
    keys = synth.splitmix64_np(np.arange(10_000 * seed, 10_000 * seed + K, dtype=np.uint64))
    res = keys[: K // 2]
    o = Oracle()
    with bmx.Engine(4 * K) as e:
        rts = rng.integers(0, 5, len(res)); rval = rng.integers(-1, 2, len(res))
        e.load_rows(res, np.full(len(res), F0), rts, rval); o.load_rows(res, np.full(len(res), F0), rts, rval)
        for b in range(3):
            mult = np.minimum(rng.geometric(0.25, K), 60)
            mult[rng.integers(0, K, 20)] = 60                      # a few long lists for sure
            idx = np.repeat(np.arange(K), mult)
            rng.shuffle(idx)
            n = len(idx)
            ts = rng.integers(0, 5, n); val = rng.integers(-1, 2, n)
            applied, flags, st = e.merge_batch(keys[idx], np.full(n, F0), ts, val, mode)
            of, ow = o.merge_batch(keys[idx], np.full(n, F0), ts, val, mode)
            assert np.array_equal(applied, ow), (seed, mode, b)
            assert (flags[applied] & 1).all()
            assert st.n_conflicts > 0 and st.n_rows == len(o)
            _assert_same_state(e, o)


def test_maximum_batch_size_boundary():
    """2^24 deltas in one call is the documented maximum (24-bit batch index in the claim tag); one more is rejected
    without touching the table. Checked against the oracle on the full batch."""
    n = 1 << 24
    R = 4_000_000
    rng = np.random.default_rng(11)
    rows = rng.integers(0, R, n)
    ids = synth.splitmix64_np((rows + 1).astype(np.uint64))
    ts = rng.integers(0, 1 << 40, n); val = rng.integers(-(1 << 30), 1 << 30, n)
    o = Oracle()
    _, ow = o.merge_batch(ids, np.full(n, F0, np.uint32), ts, val)
    with bmx.Engine(2 * R) as e:
        with pytest.raises(bmx.BmxError) as ei:
            e.merge_batch(np.concatenate([ids, ids[:1]]), np.full(n + 1, F0, np.uint32), np.concatenate([ts, ts[:1]]), np.concatenate([val, val[:1]]), want_flags=False)
        assert ei.value.code == bmx.ERR_INVALID and e.row_count() == 0
        applied, _, st = e.merge_batch(ids, np.full(n, F0, np.uint32), ts, val, want_flags=False)
        assert np.array_equal(applied, ow)
        assert st.n_rows == len(o) and rows_digest(*e.dump_rows()) == o.digest()


@pytest.mark.parametrize("name", stream_fixtures())
def test_strict_flags_equal_reference_on_every_stream(name):
    """BMX_MERGE_STRICT_FLAGS: flags[j] is exactly what resolve() returned for delta j in the reference's sequential loop,
    duplicates or not (SURVEY §8(a) batch semantics (C)); state and winners as in the default mode."""
    g = load_golden(name)
    spec = g["spec"]
    st, o, oflags, owinners = run_oracle_stream(spec)
    _, rid, rf, rts, rval = st["resident"]
    _, did, df, dts, dval = st["deltas"]
    with bmx.Engine(max(1024, 2 * (spec["R"] + spec["D"]))) as e:
        e.load_rows(rid, rf, rts, rval)
        applied, flags, stats = e.merge_batch(did, df, dts, dval, INSERT_REFERENCE | bmx.MERGE_STRICT_FLAGS)
        assert np.array_equal(flags, golden_flags(g))
        assert applied.tolist() == g["winners"]
        id, f, ts, val = e.dump_rows()
        assert "%x" % rows_digest(id, f, ts, val) == g["digest"]
        assert stats.n_rows == g["n_rows_final"] and stats.n_applied == len(g["winners"])
        # a second, default-mode batch on the same table still behaves (marks, heads and counters are consistent)
        applied2, _, _ = e.merge_batch(did, df, dts, dval, INSERT_REFERENCE)
        _, ow2 = o.merge_batch(did, df, dts, dval, INSERT_REFERENCE)     # rows created with ts := 2 are beaten again on replay
        assert np.array_equal(applied2, ow2)
        _assert_same_state(e, o)


@pytest.mark.parametrize("mode", [INSERT_REFERENCE, INSERT_DELTA])
def test_strict_flags_on_duplicate_heavy_batches(mode):
    rng = np.random.default_rng(5)
    K = 2000
    keys = synth.splitmix64_np(np.arange(777, 777 + K, dtype=np.uint64))
    o = Oracle()
    with bmx.Engine(4 * K) as e:
        rts = rng.integers(0, 5, K // 2); rval = rng.integers(-1, 2, K // 2)
        e.load_rows(keys[: K // 2], np.full(K // 2, F0), rts, rval); o.load_rows(keys[: K // 2], np.full(K // 2, F0), rts, rval)
        for b in range(3):
            mult = np.minimum(rng.geometric(0.2, K), 80)
            idx = np.repeat(np.arange(K), mult); rng.shuffle(idx)
            n = len(idx)
            ts = rng.integers(0, 5, n); val = rng.integers(-1, 2, n)
            applied, flags, st = e.merge_batch(keys[idx], np.full(n, F0), ts, val, mode | bmx.MERGE_STRICT_FLAGS)
            of, ow = o.merge_batch(keys[idx], np.full(n, F0), ts, val, mode)
            assert np.array_equal(flags, of), (mode, b)
            assert np.array_equal(applied, ow), (mode, b)
            _assert_same_state(e, o)
    with bmx.Engine(100) as e:
        with pytest.raises(bmx.BmxError):
            e.merge_batch([1], [F0], [1], [1], INSERT_REFERENCE | bmx.MERGE_STRICT_FLAGS | bmx.MERGE_UNIQUE_KEYS)


@pytest.mark.parametrize("mode", [INSERT_REFERENCE, INSERT_DELTA])
def test_same_new_key_on_every_lane_of_a_wave(mode):
    """A wave whose 64 lanes all insert the SAME absent key (one lane creates the row, 63 wait for its field), with the shared key
    starting at every lane position (the lanes before it carry distinct other keys), and a wave of 64 different fields of one new node
    (the creators' slots lie on each other's probe paths). Checked against the oracle; a mis-ordered creator/waiter pair would end in
    BMX_ERR_INTERNAL (bounded spin)."""
    rng = np.random.default_rng(5)
    o = Oracle()
    with bmx.Engine(1 << 16) as e:
        for pos in range(64):
            ids = np.concatenate([synth.splitmix64_np(np.arange(10_000 * (pos + 1), 10_000 * (pos + 1) + pos, dtype=np.uint64)),
                                  np.full(64 - pos, 7_000_000 + pos, np.uint64),
                                  np.full(64, 9_000_000 + pos, np.uint64)])
            fields = np.concatenate([np.full(64, F0, np.uint32), np.array([synth.field_hash(k) for k in range(64)], np.uint32)])
            n = len(ids)
            ts = rng.integers(0, 4, n); val = rng.integers(-2, 3, n)
            applied, _, st = e.merge_batch(ids, fields, ts, val, mode)
            _, ow = o.merge_batch(ids, fields, ts, val, mode)
            assert np.array_equal(applied, ow), (mode, pos)
        _assert_same_state(e, o)


@pytest.mark.parametrize("seed,hot", [(1, 0), (2, 40), (3, 90)])
def test_mark_created_names_the_delta_that_created_each_row(seed, hot):
    """BMX_MERGE_MARK_CREATED: bit 31 of applied_idx[k] <=> winner k took resolve()'s "no current state" branch (src/bullet-crt.js:172-185) — its row
    stores clock 2, not its own ts. Unique keys, heavy duplication (the creating delta is the SMALLEST index of a key, not whoever claimed first) and
    strict mode; every marked winner's row is read back."""
    rng = np.random.default_rng(seed)
    R, D = 3000, 20000
    rid, rf, rts, rval = synth.big_resident(R, seed=seed)
    keys = rng.integers(0, 2 * R, D)                    # half of them absent
    if hot:
        m = rng.random(D) < hot / 100
        keys[m] = 2 * R + rng.integers(0, 25, int(m.sum()))   # 25 absent hot keys
    did = streams.splitmix64_np(keys.astype(np.uint64) + np.uint64(1))
    did[keys < R] = rid[keys[keys < R]]
    df = np.full(D, rf[0], np.uint32)
    dts = rng.integers(0, 6, D).astype(np.int64) if hot else rng.integers(1, 5000, D).astype(np.int64)      # clocks around the stored 2: ties with it happen
    dval = rng.integers(-3, 4, D).astype(np.int64)
    for mode in (0, bmx.MERGE_STRICT_FLAGS):
        o = Oracle(); o.load_rows(rid, rf, rts, rval)
        want = o.merge_batch_marked(did, df, dts, dval)
        with bmx.Engine(capacity_rows=4 * (R + D)) as e:
            e.load_rows(rid, rf, rts, rval)
            applied, _, st = e.merge_batch(did, df, dts, dval, bmx.INSERT_REFERENCE | bmx.MERGE_MARK_CREATED | mode)
            assert np.array_equal(applied, want), (mode, int((applied != want).sum()) if len(applied) == len(want) else (len(applied), len(want)))
            marked = applied[(applied & bmx.APPLIED_CREATED) != 0] & bmx.APPLIED_INDEX
            assert len(marked) > 100
            ts, val, found = e.get_rows(did[marked], df[marked])
            assert found.all() and (ts == 2).all() and np.array_equal(val, dval[marked])
        o.close()
