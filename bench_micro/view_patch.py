"""The value-ordered view under writes (round 5): R-row int32 (or, with `wide`, int64) index with a view, then ROUNDS x (a 1M-delta merge on the indexed field, the first
equals after it = refresh from the change log + patch of the view, ten more equals). Prints what the caller waited for; under `rocprofv3 --kernel-trace --stats` the
per-kernel durations of the patch (k_view_keys / k_view_tile_sort / k_view_merge_pass / k_view_find / k_view_merge) come with it.
usage: python bench_micro/view_patch.py [R] [wide] [ROUNDS]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import numpy as np, torch, bmx
from bmx import synth
R = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
wide = len(sys.argv) > 2 and sys.argv[2] == "wide"
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 8
D = 1_000_000
sh = 33 if wide else 0
dev = torch.device("cuda", 0)
fa = synth.fnv1a32("n:age")
with bmx.Engine(capacity_rows=R + 1024 + (ROUNDS + 1) * D, device=0) as e:
    for r0 in range(0, R, 10_000_000):
        m = min(10_000_000, R - r0)
        ids = synth.splitmix64_np(np.arange(r0 + 1, r0 + m + 1, dtype=np.uint64))
        with np.errstate(over="ignore"):
            ages = (synth.splitmix64_np(ids ^ np.uint64(0xABCDEF)) % np.uint64(1000)).astype(np.int64)
        e.load_rows(ids, np.full(m, fa, np.uint32), np.full(m, 5, np.int64), ages << sh)
    e.index_build(fa); e.index_set_ordered(fa, 1)
    hb = bmx.HostBuffer((1 << 18) * 8); host_ids = hb.array(np.uint64, 1 << 18); host_ids[:] = 0     # the caller's page-locked answer buffer
    out_ids = torch.zeros(R // 50, dtype=torch.int64, device=dev); n_out = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    e.scan_range_dev(fa, 42 << sh, 42 << sh, out_ids, out_ids.numel(), n_out); e.sync()
    print("R = %d %s: first sort %.0f us" % (R, "int64" if wide else "int32", e.index_ordered_stats(fa)["last_sort_us"]), flush=True)
    for rnd in range(ROUNDS):
        rng = np.random.default_rng(100 + rnd)
        rows = rng.integers(0, R, D).astype(np.int64)
        rows[::10] = R + rnd * D + np.arange(len(rows[::10]))
        bid = synth.splitmix64_np((rows + 1).astype(np.uint64))
        with np.errstate(over="ignore"):
            bval = ((synth.splitmix64_np(bid ^ np.uint64(0x5151 + rnd)) % np.uint64(1000)).astype(np.int64)) << sh
        cols = (torch.from_numpy(bid.view(np.int64)).to(dev), torch.full((D,), int(np.array([fa], np.uint32).view(np.int32)[0]), dtype=torch.int32, device=dev),
                torch.full((D,), 9 + rnd, dtype=torch.int64, device=dev), torch.from_numpy(bval).to(dev))
        torch.cuda.synchronize()
        e.merge_batch_dev(D, *cols, bmx.INSERT_REFERENCE, applied=None, n_applied=n_out); e.sync()
        s0 = e.index_ordered_stats(fa)
        t0 = time.perf_counter()
        got = e.scan_range(fa, 42 << sh, 42 << sh, out=host_ids)     # host mode: returns with the ids; a rewrite of main it made due runs behind the answer
        first = (time.perf_counter() - t0) * 1e6
        t1 = time.perf_counter(); e.sync(); behind = (time.perf_counter() - t1) * 1e6
        s1 = e.index_ordered_stats(fa)
        e.sync(); e.timer_start()
        for _ in range(10):
            e.scan_range_dev(fa, 42 << sh, 42 << sh, out_ids, out_ids.numel(), n_out)
        nxt = e.timer_stop() / 10 * 1e3
        print("round %d: first equals after the merge %.0f us, work behind the answer %.0f us (patch %.0f us, %d keys, patches +%d, sorts +%d, main rewritten +%d, %d keys pending), then %.1f us per equals; %d matches" %
              (rnd, first, behind, s1["last_patch_us"], s1["keys_patched"] - s0["keys_patched"], s1["patches"] - s0["patches"], s1["sorts"] - s0["sorts"], s1["rewrites"] - s0["rewrites"], s1["pending_keys"], nxt, len(got)), flush=True)
