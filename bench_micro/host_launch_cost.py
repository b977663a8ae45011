"""Host time of one merge call (wall time inside the ctypes call, no synchronisation between calls): device-resident SoA batches and 32-byte records,
small (64: the kernels are trivial, the call's own cost shows) and 1M deltas (the device is busy: does a call wait for it?), deferral on and off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np, torch, bmx
from bmx import synth
dev = torch.device("cuda", 0)
R = 10_000_000
res = synth.big_resident(R, seed=1)
def to_dev(c):
    return (torch.from_numpy(c[0].view(np.int64)).to(dev), torch.from_numpy(c[1].view(np.int32)).to(dev), torch.from_numpy(c[2]).to(dev), torch.from_numpy(c[3]).to(dev))
with bmx.Engine(capacity_rows=2 * R + 40_000_000, device=0) as e:
    for r0 in range(0, R, 2_000_000):
        e.load_rows(*[c[r0:r0 + 2_000_000] for c in res])
    for n in (64, 100_000, 1_000_000):
        NB = 40
        bs = [to_dev(synth.big_deltas(n, R, seed=5, insert_pct=10, unique=True, batch=b, drift=1000)) for b in range(NB)]
        applied = torch.zeros(n, dtype=torch.int32, device=dev); n_applied = torch.zeros(1, dtype=torch.int64, device=dev)
        recs = torch.zeros((n, 4), dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        for defer in (True, False):
            e.set_deferred(defer)
            e.sync()
            ts = []
            for b in range(NB):
                t = time.perf_counter(); e.merge_batch_dev(n, *bs[b], 0, applied=applied, n_applied=n_applied); ts.append(time.perf_counter() - t)
            e.sync()
            ts = np.array(ts[4:]) * 1e6
            print("SoA batch of %8d, deferral %-3s: host us per call median %.1f, mean %.1f, max %.1f" % (n, "on" if defer else "off", np.median(ts), ts.mean(), ts.max()), flush=True)
        # records (the sharded receive side): partition into one slab, then merge_records
        e.set_deferred(False)
        ts = []
        for b in range(NB):
            e.partition_by_owner_slabs_dev(n, *bs[b], 1, n, recs, cnt)
            t = time.perf_counter(); e.merge_records_dev(n, recs, 0, applied=applied, n_applied=n_applied); ts.append(time.perf_counter() - t)
        e.sync()
        ts = np.array(ts[4:]) * 1e6
        print("records    of %8d              : host us per call median %.1f, mean %.1f, max %.1f" % (n, np.median(ts), ts.mean(), ts.max()), flush=True)
