"""Same-process A/B of deferred compaction (include/bmx.h): ONE table, ONE stream of batches; rounds of K steps alternate between
compaction on the merge stream (A: K1 K2 K3 | K1 K2 K3 ...) and compaction under the next probe kernel (B: K1 K2 | K1 K2 ..., K3 on the side
stream). Box spread cannot hide the difference: both arms run on the same table minutes apart at most. us per step between HIP events.
usage: defer_ab.py [config 2|5] [rounds] [steps per round]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np, torch, bmx
from bmx import synth
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
R, D, DT = 10_000_000, 1_000_000, 1_000_000
dev = torch.device("cuda", 0)
def to_dev(c):
    i, f, t, v = c
    return (torch.from_numpy(i.view(np.int64)).to(dev), torch.from_numpy(f.view(np.int32)).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(v).to(dev))
def gen(b):
    if cfg == 5:
        return synth.big_deltas(D, R, seed=52, insert_pct=0, hot_pct=30, hot_keys=R // 1000, unique=False, batch=b, drift=DT // 2)
    return synth.big_deltas(D, R, seed=2, insert_pct=10, unique=True, batch=b, drift=DT // 16)
e = bmx.Engine(22_000_000 + ROUNDS * K * (D // 10))
e.load_rows(*synth.big_resident(R, seed=1))
applied = torch.zeros((K, D), dtype=torch.int32, device=dev); na = torch.zeros(K, dtype=torch.int64, device=dev)
b = 0
for w in range(3):
    e.merge_batch_dev(D, *to_dev(gen(b)), bmx.INSERT_REFERENCE, applied=applied[0], n_applied=na[0:1]); b += 1
e.sync()
res = {False: [], True: []}
for rnd in range(ROUNDS):
    for arm in ((False, True) if rnd % 2 == 0 else (True, False)):
        bs = [to_dev(gen(b + i)) for i in range(K)]; b += K
        torch.cuda.synchronize()
        e.set_deferred(arm)
        e.sync(); e.timer_start()
        for i in range(K):
            e.merge_batch_dev(D, *bs[i], bmx.INSERT_REFERENCE, applied=applied[i], n_applied=na[i:i + 1])
        us = e.timer_stop() / K * 1e3
        res[arm].append(us)
        print("round %d %s: %.2f us/step (winners/step %.0f)" % (rnd, "deferred " if arm else "in-stream", us, float(na.float().mean().item())), flush=True)
a, d = np.array(res[False]), np.array(res[True])
print("config %d, %d rounds x %d steps: in-stream median %.2f (min %.2f), deferred median %.2f (min %.2f): %+.2f us/step; deferred/side counts %s" %
      (cfg, ROUNDS, K, np.median(a), a.min(), np.median(d), d.min(), np.median(d) - np.median(a), e.deferred_counts()))
e.close()
