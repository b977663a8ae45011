"""Seeded synthetic delta streams — TEST INFRASTRUCTURE ONLY.

gen_stream() mirrors genStream() of oracle/gen_golden.js bit for bit (xorshift32, same draw order), so
the golden fixtures only need to store the spec, not the inputs. (Bench-sized synthetic inputs come from
the product-side generator bmx/synth.py, which is plain data generation and not part of the oracle.)
"""
import numpy as np

PERM_PRIME = 1000003
M64 = (1 << 64) - 1


def splitmix64(x):
    z = (x + 0x9e3779b97f4a7c15) & M64
    z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & M64
    z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & M64
    return z ^ (z >> 31)


def splitmix64_np(x):
    with np.errstate(over="ignore"):
        z = x.astype(np.uint64) + np.uint64(0x9e3779b97f4a7c15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
        return z ^ (z >> np.uint64(31))


def fnv1a32(s):
    h = 0x811c9dc5
    for ch in s.encode("utf-8"):
        h ^= ch
        h = (h * 0x01000193) & 0xffffffff
    return h


def field_hash(fi):
    return fnv1a32("f%d" % fi)


class XorShift32:
    def __init__(self, seed):
        self.s = (seed & 0xffffffff) or 0x9e3779b9

    def __call__(self):
        s = self.s
        s ^= (s << 13) & 0xffffffff
        s ^= s >> 17
        s ^= (s << 5) & 0xffffffff
        self.s = s
        return s


def row_id(row, F):
    return splitmix64(row // F + 1)


def rows_to_keys(rows, F):
    rows = np.asarray(rows, dtype=np.int64)
    ids = splitmix64_np((rows // F + 1).astype(np.uint64))
    fh = np.array([field_hash(i) for i in range(F)], dtype=np.uint32)
    return ids, fh[rows % F]


def gen_stream(spec):
    """Returns dict(resident=(rows, id, field, ts, val), deltas=(rows, id, field, ts, val))."""
    rng = XorShift32(spec["seed"])
    F = spec.get("F", 1) or 1
    R, D = spec["R"], spec["D"]
    T0, DT, VR, VOFF = spec["T0"], spec["DT"], spec["VR"], spec["VOFF"]
    rts = np.zeros(R, np.int64); rval = np.zeros(R, np.int64)
    for r in range(R):
        rts[r] = T0 + rng() % DT
        rval[r] = rng() % VR - VOFF
    ins_space = spec.get("ins_space") or max(1, R // 10)
    drow = np.zeros(D, np.int64); dts = np.zeros(D, np.int64); dval = np.zeros(D, np.int64)
    for j in range(D):
        u = rng() % 100
        if u < spec["insert_pct"]:
            row = R + j if spec["unique"] else R + rng() % ins_space
        elif u < spec["insert_pct"] + spec["hot_pct"]:
            row = rng() % spec["H"]
        elif spec["unique"]:
            row = (j * PERM_PRIME + 7) % R
        else:
            row = rng() % R
        drow[j] = row
        dts[j] = T0 + rng() % (2 * DT)
        dval[j] = rng() % VR - VOFF
    rrow = np.arange(R, dtype=np.int64)
    rid, rf = rows_to_keys(rrow, F)
    did, df = rows_to_keys(drow, F)
    return dict(F=F, resident=(rrow, rid, rf, rts, rval), deltas=(drow, did, df, dts, dval))
