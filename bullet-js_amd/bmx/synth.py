"""synth.py — synthetic graphs and delta streams (SURVEY §8(d) config 2/4/5 shapes). Data generation only:
counter-based (vectorised numpy), identical on every rank, no reference or oracle code involved."""
import numpy as np

PERM_PRIME = 1000003
M64 = (1 << 64) - 1


def splitmix64_np(x):
    with np.errstate(over="ignore"):
        z = np.asarray(x).astype(np.uint64) + np.uint64(0x9e3779b97f4a7c15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
        return z ^ (z >> np.uint64(31))


def fnv1a32(s):
    """32-bit FNV-1a of a UTF-8 string: the host-side field-name hash (0xFFFFFFFF is remapped: reserved)."""
    h = 0x811c9dc5
    for ch in s.encode("utf-8"):
        h ^= ch
        h = (h * 0x01000193) & 0xffffffff
    return h if h != 0xffffffff else 0xfffffffe


def field_hash(fi):
    return fnv1a32("f%d" % fi)


def rows_to_keys(rows, F=1):
    """row ordinal -> (node id hash u64, field hash u32); F fields per node."""
    rows = np.asarray(rows, dtype=np.int64)
    ids = splitmix64_np((rows // F + 1).astype(np.uint64))
    fh = np.array([field_hash(i) for i in range(F)], dtype=np.uint32)
    return ids, fh[rows % F]


def _mix64_np(x):
    with np.errstate(over="ignore"):
        x = x ^ (x >> np.uint64(33)); x = x * np.uint64(0xff51afd7ed558ccd)
        x = x ^ (x >> np.uint64(33)); x = x * np.uint64(0xc4ceb9fe1a85ec53)
        return x ^ (x >> np.uint64(33))


def owner_of_np(ids, nshards):
    """numpy twin of bmx_owner_of() (include/bmx.h): floor(owner_hash(id) * nshards / 2^64)."""
    ids = np.asarray(ids, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _mix64_np(ids * np.uint64(0xD6E8FEB86659FD93) + np.uint64(0x2545F4914F6CDD1D))
    # 64x64 -> high 64 via two 32-bit halves (nshards < 2^32)
    g = np.uint64(nshards)
    lo = (h & np.uint64(0xffffffff)) * g
    hi = (h >> np.uint64(32)) * g
    return ((hi + (lo >> np.uint64(32))) >> np.uint64(32)).astype(np.uint32)


def _u(seed, n, salt):
    """n uniform uint64 draws, counter-based (vectorised)."""
    with np.errstate(over="ignore"):
        i = np.arange(n, dtype=np.uint64)
        return splitmix64_np(i * np.uint64(0x9E3779B97F4A7C15) + np.uint64((seed * 0x632BE59BD9B4E019 + salt * 0xD1342543DE82EF95) & M64))


def big_resident(R, seed=1, T0=1_000_000, DT=1_000_000, F=1, row0=0):
    """Resident rows row0..row0+R-1: id = splitmix64(node+1), ts~U[T0,T0+DT), val in ±2^31."""
    rows = np.arange(row0, row0 + R, dtype=np.int64)
    ids, fld = rows_to_keys(rows, F)
    ts = (T0 + (_u(seed, R, 1) % np.uint64(DT))).astype(np.int64)
    val = (_u(seed, R, 2) % np.uint64(1 << 32)).astype(np.int64) - (1 << 31)
    return ids, fld, ts, val


def big_deltas(D, R, seed=2, T0=1_000_000, DT=1_000_000, F=1, insert_pct=10, hot_pct=0, hot_keys=0, unique=True, batch=0, drift=None, part=(0, 1)):
    """Config-2/5 shaped delta batch over a resident graph of R rows.

    unique=True: hit rows are a stride permutation (no duplicate keys inside the batch), inserts get fresh rows.
    ts ~ U[T0 + batch*drift, T0 + batch*drift + 2*DT); drift defaults to DT/2 per batch (streaming, config 5).
    Config 2 uses drift = DT/16: consecutive unique batches walk disjoint rows (a row is revisited every R/D batches),
    and the slow drift keeps ~75-78 % of hits winning in steady state, as SURVEY §8(d) specifies.
    part=(rank, world): with unique=True the originators of one global step walk DISJOINT rows (and insert disjoint new rows), so the
    union of the world's batches of a step has unique keys too; timestamps still follow `batch`."""
    if drift is None:
        drift = DT // 2
    u = _u(seed + 7919 * batch, D, 3) % np.uint64(100)
    j = np.arange(D, dtype=np.int64)
    if unique:
        gb = batch * int(part[1]) + int(part[0])          # position of this batch in the global stream of unique batches
        hit_rows = ((j + gb * D) * PERM_PRIME + 7) % R
        ins_rows = R + gb * D + j
    else:
        hit_rows = (_u(seed + 7919 * batch, D, 4) % np.uint64(R)).astype(np.int64)
        ins_rows = R + (_u(seed + 7919 * batch, D, 5) % np.uint64(max(1, R // 10))).astype(np.int64)
    rows = np.where(u < insert_pct, ins_rows, hit_rows)
    if hot_pct:
        hot_rows = (_u(seed + 7919 * batch, D, 6) % np.uint64(max(1, hot_keys))).astype(np.int64)
        rows = np.where((u >= insert_pct) & (u < insert_pct + hot_pct), hot_rows, rows)
    ids, fld = rows_to_keys(rows, F)
    ts = (T0 + batch * drift + (_u(seed + 7919 * batch, D, 8) % np.uint64(2 * DT))).astype(np.int64)
    val = (_u(seed + 7919 * batch, D, 9) % np.uint64(1 << 32)).astype(np.int64) - (1 << 31)
    return ids, fld, ts, val
