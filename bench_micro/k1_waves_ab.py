"""Same-process A/B of the probe kernel's occupancy cap (BMX_K1_WAVES: resident waves per SIMD): ONE table, engines are not re-created — the cap is read per
context at creation, so one context per arm over the same device memory budget; per arm: K1 alone (per-kernel HIP events) and the deferred step. config 2 / 5."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np, torch, bmx
from bmx import synth
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
arms = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8,6,5,4,3").split(",")]
R, D, DT, K = 10_000_000, 1_000_000, 1_000_000, 20
dev = torch.device("cuda", 0)
def to_dev(c):
    i, f, t, v = c
    return (torch.from_numpy(i.view(np.int64)).to(dev), torch.from_numpy(f.view(np.int32)).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(v).to(dev))
def gen(b):
    if cfg == 5:
        return synth.big_deltas(D, R, seed=52, insert_pct=0, hot_pct=30, hot_keys=R // 1000, unique=False, batch=b, drift=DT // 2)
    return synth.big_deltas(D, R, seed=2, insert_pct=10, unique=True, batch=b, drift=DT // 16)
res = synth.big_resident(R, seed=1)
bs = [to_dev(gen(b)) for b in range(3 + 2 * K)]
applied = torch.zeros((K, D), dtype=torch.int32, device=dev); na = torch.zeros(K, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
for rnd in range(2):
    for w in arms:
        os.environ["BMX_K1_WAVES"] = str(w)
        e = bmx.Engine(22_000_000 + 50 * (D // 10)); e.load_rows(*res)
        for b in range(3): e.merge_batch_dev(D, *bs[b], bmx.INSERT_REFERENCE, applied=applied[0], n_applied=na[0:1])
        e.sync(); e.timer_start()
        for i in range(K): e.merge_batch_dev(D, *bs[3 + i], bmx.INSERT_REFERENCE, applied=applied[i], n_applied=na[i:i + 1])
        step = e.timer_stop() / K * 1e3
        e.profile_enable(True)
        for i in range(12): e.merge_batch_dev(D, *bs[3 + K + i], bmx.INSERT_REFERENCE, applied=applied[0], n_applied=na[0:1])
        ms, n = e.profile_read(); e.profile_enable(False)
        print("round %d waves/SIMD %d: deferred step %.2f us; alone: K1 %.2f K2 %.2f K3 %.2f us" % (rnd, w, step, ms["probe_apply"] * 1e3, ms["resolve_lists"] * 1e3, ms["compact"] * 1e3), flush=True)
        e.close()
