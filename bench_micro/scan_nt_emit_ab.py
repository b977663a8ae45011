"""Same-process A/B of the id output's memory policies (round 5): BMX_SCAN_NT = 0 (round 4), 1 (gathered ids loaded nontemporally),
2 (output stored nontemporally), 3 (both), at 1 / 10 / 50 % selectivity, int32 index of R rows; us per whole scan (mask + emit), back to back between HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np, torch, bmx
from bmx import synth
R = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ARMS = tuple(sys.argv[2].split(",")) if len(sys.argv) > 2 else ("0", "1", "2", "3")       # bit 2 (4): streamed blocks keep 16 loads per lane in flight instead of 8
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
    out_ids = torch.zeros(R, dtype=torch.int64, device=dev); n_out = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    for name, lo, hi in [("1%", 100, 109), ("10%", 100, 199), ("30%", 0, 299), ("50%", 0, 499), ("100%", 0, 999)]:
        res, chk = {}, set()
        for rnd in range(3):
            for arm in ARMS:
                os.environ["BMX_SCAN_NT"] = arm
                for _ in range(3): e.scan_range_dev(fa, lo, hi, out_ids, R, n_out)
                e.sync(); e.timer_start()
                for _ in range(10): e.scan_range_dev(fa, lo, hi, out_ids, R, n_out)
                res.setdefault(arm, []).append(e.timer_stop() / 10 * 1e3)
                chk.add(int(out_ids[:int(n_out.item())].sum().item()))
        assert len(chk) == 1
        print("%s rows, %s: " % (R, name) + " | ".join("nt=%s %s" % (a, " ".join("%.1f" % x for x in res[a])) for a in ARMS), flush=True)
