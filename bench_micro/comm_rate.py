"""Side measurement for DESIGN.md: N logical shards on ONE GPU behind bmx_comm_* (what a one-GPU box can say about the
one-process multi-shard entry): device-resident steps (every shard originates D/N deltas, owner partitions scatter straight into the
owners' receive slabs, every shard merges) and host batches. Not a scaling figure: all shards share one GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np
import torch
import bmx
from bmx import synth
R, D, NB = 10_000_000, 1_000_000, 10
dev = torch.device("cuda", 0)
def to_dev(c):
    i, f, t, v = c
    return (torch.from_numpy(i.view(np.int64)).to(dev), torch.from_numpy(f.view(np.int32)).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(v).to(dev))
res = synth.big_resident(R, seed=1)
for N in (1, 2, 4, 8):
    with bmx.Comm([0] * N, capacity_rows_per_shard=22_000_000 // N + 4096) as c:
        c.load_rows(*res)
        steps = []
        for b in range(NB):
            steps.append([(D // N,) + to_dev(synth.big_deltas(D // N, R, seed=2, insert_pct=10, unique=True, batch=b, drift=62500, part=(i, N))) for i in range(N)])
        c.merge_dev(steps[0]); c.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in steps[1:]:
            c.merge_dev(s)
        c.sync()
        dt = (time.perf_counter() - t0) / (NB - 1)
        hb = [synth.big_deltas(D, R, seed=7, insert_pct=10, unique=True, batch=20 + b, drift=62500) for b in range(8)]
        keep, *pin = bmx.host_columns(D)          # the caller's arrays in page-locked memory (bmx_host_alloc), refilled per batch like a host that reuses its buffers
        keep_out = bmx.HostBuffer(4 * D); out = keep_out.array(np.uint32, D)
        def fill(h):
            for d, s in zip(pin, h):
                d[:] = s
        fill(hb[0]); c.merge(*pin, applied_out=out)
        ts = []
        for h in hb[1:]:
            fill(h)
            t0 = time.perf_counter()
            c.merge(*pin, applied_out=out)
            ts.append(time.perf_counter() - t0)
        dth = sorted(ts)[len(ts) // 2]
        print("N=%d logical shards on one GPU: device step (1M deltas in all) %.0f us = %.2f G merges/s | host batch of 1M (page-locked arrays) %.0f us = %.2f G/s" % (N, dt * 1e6, D / dt / 1e9, dth * 1e6, D / dth / 1e9))
        keep.close(); keep_out.close()
