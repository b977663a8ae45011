"""Side experiment (not bench.py): PATHOLOGICAL skew — SKEW deltas of a 1M-delta batch hit ONE key, the rest is config-2 shaped. The state is compared
with the oracle. profiles/r03_skew.log holds this script's output from the last commit that still had the bucketed merge path (rounds 1-2,
BMX_MERGE_BUCKETED, csrc/bin_kernels.h), run on both paths: the path that was kept "for pathological skew" took 0.37 / 3.0 / 29.9 ms per step at
10^3 / 10^4 / 10^5 deltas on one key, the default path 99-106 us (a delta below the value it sees drops out without claiming, so a hot row's list holds
the running maxima only). The bucketed path was deleted on that evidence."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np, torch
import bmx
from bmx import synth
from oracle.oracle import Oracle, rows_digest

R, D, NB = 10_000_000, 1_000_000, 5
dev = torch.device("cuda", 0)
def to_dev(c): return (torch.from_numpy(c[0].view(np.int64)).to(dev), torch.from_numpy(c[1].view(np.int32)).to(dev), torch.from_numpy(c[2]).to(dev), torch.from_numpy(c[3]).to(dev))
res = synth.big_resident(R, seed=1)
for skew in [int(x) for x in os.environ.get("SKEW", "1000,10000,100000").split(",")]:
    batches = []
    rng = np.random.default_rng(skew)
    for b in range(NB):
        i, f, t, v = [np.array(x) for x in synth.big_deltas(D, R, seed=9, insert_pct=10, unique=True, batch=b)]
        hot = rng.choice(D, skew, replace=False)
        i[hot] = res[0][12345]; f[hot] = res[1][12345]            # one resident key
        t[hot] = rng.integers(1, 1 << 40, skew); v[hot] = rng.integers(-1000, 1000, skew)
        batches.append((i, f, t, v))
    o = Oracle(); o.load_rows(*res)
    for b in batches: o.merge_batch(*b)
    e = bmx.Engine(22_000_000); e.load_rows(*res)
    dd = [to_dev(b) for b in batches]
    applied = torch.zeros(D, dtype=torch.int32, device=dev); n_applied = torch.zeros(NB, dtype=torch.int64, device=dev)
    e.merge_batch_dev(D, *dd[0], bmx.INSERT_REFERENCE, applied=applied, n_applied=n_applied[0:1])
    e.sync(); e.profile_enable(True); e.timer_start()
    for b in range(1, NB): e.merge_batch_dev(D, *dd[b], bmx.INSERT_REFERENCE, applied=applied, n_applied=n_applied[b:b + 1])
    ms = e.timer_stop(); st, n = e.profile_read(); e.profile_enable(False)
    ok = rows_digest(*e.dump_rows()) == o.digest()
    print("skew %6d deltas on one key | %9.1f us/step  stages(us) %s  state == oracle: %s" % (skew, ms / (NB - 1) * 1e3, {k: round(x * 1e3, 1) for k, x in st.items()}, ok), flush=True)
    e.close(); o.close()
