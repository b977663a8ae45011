"""GPU parity at BASELINE.json's full sizes: the benched shapes themselves, checked against the oracle.

config 2: 10M-row resident graph, 1M-delta batches (90 % hits on unique keys, 10 % inserts) — winners, n_rows and the state digest;
config 5: the same graph under 10 streaming batches of 1M deltas with 30 % of them on R/1000 hot keys.
The oracle side is computed once per shape."""
import functools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import bmx
from bmx import synth
from oracle.oracle import Oracle, rows_digest

R, D = 10_000_000, 1_000_000
T0 = DT = 1_000_000


def _batches(shape):
    if shape == "config2":
        return [synth.big_deltas(D, R, seed=2, T0=T0, DT=DT, insert_pct=10, unique=True, batch=b, drift=DT // 16) for b in range(3)]
    return [synth.big_deltas(D, R, seed=5, T0=T0, DT=DT, insert_pct=0, hot_pct=30, hot_keys=R // 1000, unique=False, batch=b, drift=DT // 2) for b in range(10)]


@functools.lru_cache(maxsize=2)
def _oracle_side(shape):
    res = synth.big_resident(R, seed=1, T0=T0, DT=DT)
    o = Oracle()
    o.load_rows(*res)
    bs = _batches(shape)
    winners = [o.merge_batch(*b)[1] for b in bs]
    out = (res, bs, winners, len(o), o.digest())
    o.close()
    return out


@pytest.mark.parametrize("shape", ["config2", "config5"])
def test_benched_shape_matches_oracle(shape):
    res, bs, winners, n_rows, digest = _oracle_side(shape)
    with bmx.Engine(22_000_000) as e:
        e.load_rows(*res)
        for b, want in zip(bs, winners):
            applied, _, st = e.merge_batch(*b, want_flags=False)
            assert np.array_equal(applied, want), (shape, len(applied), len(want))
            if shape == "config5":
                assert st.n_conflicts > 0
        assert e.row_count() == n_rows
        assert rows_digest(*e.dump_rows()) == digest
