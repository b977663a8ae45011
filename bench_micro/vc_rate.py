"""N4 rate probe: 10M resident K=3 rows, 1M-delta batches (10% new keys) through bmx_vc_merge_batch (host buffers).
Run under `rocprofv3 --kernel-trace --stats` for the k_vc_link / k_vc_resolve durations; prints the PCIe-inclusive rate."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", "bullet-js_amd"))
import bmx

R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
K, local = 3, 2
rng = np.random.default_rng(1)
mix = lambda x: (x.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(0x1234567)
e = bmx.EngineVC(R + 8 * D, K, local)
ids = mix(np.arange(R)); f = np.full(R, 7, np.uint32)
e.load_rows(ids, f, rng.integers(1, 4, (R, K)).astype(np.uint32), rng.integers(0, 100, R))
print("rows", e.row_count(), flush=True)
for b in range(6):
    rows = rng.integers(0, R, D); new = rng.random(D) < 0.1
    rows[new] = R + b * D + np.arange(int(new.sum()))
    did = mix(rows); clocks = rng.integers(1, 5 + b, (D, K)).astype(np.uint32); val = rng.integers(0, 100, D)
    t0 = time.perf_counter()
    fl, upd = e.merge_batch(did, np.full(D, 7, np.uint32), clocks, val)
    dt = time.perf_counter() - t0
    print("batch %d: %.2f ms host-inclusive (%.1f M deltas/s), updated %d, concurrent %d" % (b, dt * 1e3, D / dt / 1e6, len(upd), int((fl & 8).astype(bool).sum())), flush=True)
# device-pointer form: inputs resident, REPS batches back to back
import torch
dev = torch.device("cuda", 0)
REPS = 10
batches = []
for b in range(REPS):
    rows = rng.integers(0, R, D); new = rng.random(D) < 0.1
    rows[new] = R + (6 + b) * D + np.arange(int(new.sum()))
    batches.append([torch.from_numpy(x).to(dev) for x in (mix(rows).view(np.int64), np.full(D, 7, np.int32), rng.integers(1, 12, (D, K)).astype(np.int32).reshape(-1), rng.integers(0, 100, D))])
upd = torch.zeros(D, dtype=torch.int32, device=dev); nu = torch.zeros(1, dtype=torch.int64, device=dev)
e.merge_batch_dev(D, *batches[0], updated=upd, n_updated=nu); e.sync()
t0 = time.perf_counter()
for b in range(1, REPS):
    e.merge_batch_dev(D, *batches[b], updated=upd, n_updated=nu)
e.sync()
dt = (time.perf_counter() - t0) / (REPS - 1)
print("device-resident: %.1f us per 1M-delta batch = %.2f G clock-merges/s" % (dt * 1e6, D / dt / 1e9), flush=True)
