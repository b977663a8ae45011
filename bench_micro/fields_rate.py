"""Multi-field workload: R rows = R/F nodes with F integer fields each; a batch carries whole nodes (the F rows of a node are
adjacent, as a sync chunk of objects is). Prints the per-kernel times of the merge. BMX_LIB_PATH selects the build."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bullet-js_amd"))
import bmx
from bmx import synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
R, D, NB = 10_000_000, 1_000_000, 10
dev = torch.device("cuda", 0)
e = bmx.Engine(22_000_000)
e.load_rows(*synth.big_resident(R, seed=1, F=F))
rng = np.random.default_rng(3)
nodes = R // F
batches = []
for b in range(NB):
    nd = rng.permutation(nodes)[: D // F].astype(np.int64)            # distinct resident nodes, all their fields
    new = rng.random(len(nd)) < 0.1
    nd[new] = nodes + b * (D // F) + np.arange(int(new.sum()))        # 10 % new nodes
    rows = (nd[:, None] * F + np.arange(F)[None, :]).reshape(-1)
    ids, fld = synth.rows_to_keys(rows, F)
    ts = rng.integers(1_000_000 + b * 62500, 3_000_000 + b * 62500, len(rows)).astype(np.int64)
    val = rng.integers(-2**31, 2**31, len(rows)).astype(np.int64)
    batches.append([torch.from_numpy(x).to(dev) for x in (ids.view(np.int64), fld.view(np.int32), ts, val)])
n = len(rows)
applied = torch.zeros(n, dtype=torch.int32, device=dev); na = torch.zeros(1, dtype=torch.int64, device=dev)
for b in range(2):
    e.merge_batch_dev(n, *batches[b], bmx.INSERT_REFERENCE, applied=applied, n_applied=na)
e.sync(); e.profile_enable(True); e.timer_start()
for b in range(2, NB):
    e.merge_batch_dev(n, *batches[b], bmx.INSERT_REFERENCE, applied=applied, n_applied=na)
ms = e.timer_stop(); st, k = e.profile_read(); e.profile_enable(False)
print("F=%d lib=%s: %.1f us/step  %s" % (F, os.path.basename(bmx.LIB_PATH), ms / (NB - 2) * 1e3, {a: round(v * 1e3, 1) for a, v in st.items()}), flush=True)
