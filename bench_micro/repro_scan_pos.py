import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/bullet-js_amd")
import numpy as np, torch, bmx
from bmx import synth
R = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
fa = synth.fnv1a32("n:age")
with bmx.Engine(capacity_rows=R + 1024, device=0) as e:
    for r0 in range(0, R, 10_000_000):
        m = min(10_000_000, R - r0)
        ids = synth.splitmix64_np(np.arange(r0 + 1, r0 + m + 1, dtype=np.uint64))
        with np.errstate(over="ignore"):
            ages = (synth.splitmix64_np(ids ^ np.uint64(0xABCDEF)) % np.uint64(1000)).astype(np.int64)
        e.load_rows(ids, np.full(m, fa, np.uint32), np.full(m, 5, np.int64), ages)
    e.index_build(fa)
    out_ids = torch.zeros(R, dtype=torch.int64, device=dev); out_pos = torch.zeros(R, dtype=torch.int32, device=dev); id_col = torch.zeros(R, dtype=torch.int64, device=dev)
    n_out = torch.zeros(1, dtype=torch.int64, device=dev)
    e.index_ids_dev(fa, 0, R, id_col); e.sync(); torch.cuda.synchronize()
    print("id_col nonzero:", int((id_col != 0).sum().item()), "of", R)
    for lo, hi in [(42, 42), (100, 109), (100, 199)]:
        e.scan_range_dev(fa, lo, hi, out_ids, R, n_out); e.sync(); m = int(n_out.item())
        e.scan_range_pos_dev(fa, lo, hi, out_pos, R, n_out); e.sync(); mp = int(n_out.item())
        a = out_ids[:m]; b = id_col[out_pos[:mp].long()]
        bad = (a != b).nonzero().flatten() if m == mp else None
        print((lo, hi), "m", m, "mp", mp, "mismatches", None if bad is None else int(bad.numel()), "first", None if bad is None or bad.numel() == 0 else (int(bad[0]), int(out_pos[bad[0]]), hex(int(a[bad[0]]) & (2**64-1)), hex(int(b[bad[0]]) & (2**64-1))))
        if bad is not None and bad.numel():
            p = out_pos[:mp].long(); print("  positions sorted ascending:", bool((p[1:] > p[:-1]).all().item()), "max pos", int(p.max()), "min", int(p.min()))
