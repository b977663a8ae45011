"""Shared helpers for the parity tests (oracle side + golden loading)."""
import base64
import glob
import json
import os

import numpy as np

from oracle import streams
from oracle.oracle import Oracle, INSERT_REFERENCE

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def stream_fixtures():
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "g2_stream_*.json")))


def golden_flags(g):
    return np.frombuffer(base64.b64decode(g["flags_b64"]), dtype=np.uint8)


def run_oracle_stream(spec, insert_mode=INSERT_REFERENCE):
    st = streams.gen_stream(spec)
    o = Oracle()
    _, rid, rf, rts, rval = st["resident"]
    o.load_rows(rid, rf, rts, rval)
    _, did, df, dts, dval = st["deltas"]
    flags, winners = o.merge_batch(did, df, dts, dval, insert_mode)
    return st, o, flags, winners


def final_rows_by_ordinal(st, o):
    """[(row ordinal, ts, val)] sorted by ordinal, like the fixture's final_rows."""
    F = st["F"]
    rows = np.unique(np.concatenate([st["resident"][0], st["deltas"][0]]))
    ids, fld = streams.rows_to_keys(rows, F)
    out = []
    for r, i, f in zip(rows.tolist(), ids.tolist(), fld.tolist()):
        got = o.get_row(i, f)
        if got is not None:
            out.append([r, got[0], got[1]])
    return out
