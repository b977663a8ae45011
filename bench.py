#!/usr/bin/env python3
"""bench.py — CRDT field-merges/s on MI355X (BASELINE.json metric), one JSON line on rank 0.

A "step" is one pass of the merge hot path over one 1M-delta batch per GPU:
  --config 2 (default; configs[1], and configs[3] when launched on N > 1 GPUs)
      resident graph R = 10M rows per GPU (id = splitmix64(node), 1 field, ts ~ U[T0,T0+DT), val in +-2^31),
      batch D = 1M deltas: 90 % hit resident rows (unique keys inside the batch), 10 % absent keys (inserts),
      ts ~ U[T0 + b*DT/16, T0 + b*DT/16 + 2*DT)  (~75-78 % of hits win in steady state; batches walk disjoint rows).
  --config 5 (configs[4]: streaming sync replay)
      same graph, 100 batches of 1M deltas per GPU: 30 % of a batch on a hot set of R/1000 keys, 70 % uniform, no inserts,
      ts ~ U[T0 + b*DT/2, T0 + b*DT/2 + 2*DT); default --warmup 10 --steps 90 = steady state over batches 10..99.
Inputs and outputs are device-resident when the timed region starts; the timed region is exactly K bmx_merge_batch calls
(BMX_MEM_DEVICE: probe+apply, conflict resolve, winner compaction) per rank.

N > 1 (one process per GPU, launched by torch.distributed.run): the graph is sharded by node-id hash, every rank originates
1M mixed-owner deltas per step; a step = partition by owner + all-to-all (RCCL) + local merge of what arrived. Weak scaling.

After the timed region (never inside it) the run CHECKS ITSELF: every timed batch's winner list, the row count and an
order-independent digest of the whole device table are compared with the CPU oracle replaying the same batches; a mismatch
fails the run. Besides the contract fields the line carries
  roofline     — dominant kernel (k_probe_apply): algorithmic bytes per launch / its average duration, measured live with
                 HIP events on the engine's stream in a second pass over fresh batches of the same shape;
  scan_config3 — configs[2]: equals/range scans over an indexed int32 field and a wide int64 field at 10M and 100M rows, with the roofline of the
                 mask kernel (the one read of the value column);
  cpu_baseline — the CPU oracle (oracle/bmx_oracle.c, proven equal to the reference on golden vectors) timed on one host
                 core on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "bullet-js_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

R_PER_GPU = 10_000_000
D_PER_STEP = int(os.environ.get("BMX_BENCH_DELTAS", 1_000_000))   # BASELINE configs: 1M; the override is for host-bound experiments only
T0, DT = 1_000_000, 1_000_000
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
CONFIG = 2             # set by --config


def say(msg, rank=None):
    """progress line on stderr, stamped with the seconds since the launcher started (or since this process did): what a killed run leaves behind"""
    t0 = float(os.environ.get("BMX_BENCH_LAUNCHED_AT", 0) or 0) or _T_IMPORT
    r = os.environ.get("RANK", "0") if rank is None else rank
    print("bench[rank %s] +%.1fs: %s" % (r, time.time() - t0, msg), file=sys.stderr, flush=True)


_T_IMPORT = time.time()


def gen_resident(R, row0=0):
    from bmx import synth
    return synth.big_resident(R, seed=1, T0=T0, DT=DT, row0=row0)


def gen_batch(b, R, D=None, seed=2, part=(0, 1)):
    from bmx import synth
    D = D_PER_STEP if D is None else D
    if CONFIG == 5:    # 30 % of the deltas on R/1000 hot keys, the rest uniform over the graph; streaming drift DT/2 per batch
        return synth.big_deltas(D, R, seed=seed + 50, part=part, T0=T0, DT=DT, insert_pct=0, hot_pct=30, hot_keys=max(1, R // 1000), unique=False, batch=b, drift=DT // 2)
    return synth.big_deltas(D, R, seed=seed, part=part, T0=T0, DT=DT, insert_pct=int(os.environ.get("BMX_BENCH_INSERT_PCT", 10)), unique=True, batch=b, drift=DT // 16)


def to_dev(cols, dev):
    id, field, ts, val = cols
    return (torch.from_numpy(id.view(np.int64)).to(dev), torch.from_numpy(field.view(np.int32)).to(dev),
            torch.from_numpy(ts).to(dev), torch.from_numpy(val).to(dev))


def exchange_selftest(dev, dist, rank, world):
    """N>1 only, before the real graph is built: a small graph (20k rows per rank) and six batches of 8192 deltas per rank go through the very pipeline the
    timed run uses — direct exchange if every rank can set it up — and every rank compares its shard with the oracle. The direct exchange stores into
    other GPUs' memory; if the data that arrives is not what was sent on THIS machine — or any rank fails any step of it: an IPC open, the set-up,
    a device-side wait that expires (60 s), an exception of any kind — every rank switches to the RCCL all-to-all for the timed run, inside the same
    processes, instead of finding out in the post-run verification. -> (exchange kind to use: "direct" / "rccl", why direct was refused or None)"""
    import bmx
    from bmx import synth
    from bmx.sharded import ShardedGraph, EngineOps
    from oracle.oracle import Oracle, rows_digest
    Rs, Ds, NBs = 20_000, 8192, 6
    kind, ok, why = "rccl", True, None
    e = sg = o = None
    try:      # ANY failure on ANY rank (an IPC open, a set-up step, a device-side wait that expired, wrong data) must end in the agreement below, never in a rank that left
        e = bmx.Engine(capacity_rows=4 * (Rs + NBs * Ds), device=dev.index or 0)
        e.set_wait_limit(WAIT_LIMIT_S)           # a peer that never stores its arrival word costs this long, once (well inside COLL_TIMEOUT_S)
        sg = ShardedGraph(EngineOps(e, dev), dist, rank, world)
        sg.load_owned_resident(Rs, T0=T0, DT=DT)
        Rg = Rs * world
        gen = lambda b, src: synth.big_deltas(Ds, Rg, seed=77 + 1000 * src, part=(src, world), T0=T0, DT=DT, insert_pct=10, unique=True, batch=b, drift=DT // 16)
        sg.setup_pipeline(Ds, slack=1.5)
        kind = sg.exchange
        if kind != "direct":
            why = getattr(sg, "direct_refused", None) or "a rank could not set up or verify the IPC mappings"
    except Exception as err:
        ok, why = False, "self-test set-up raised on rank %d: %s" % (rank, str(err)[:200])
    if kind == "direct" and ok:
        try:
            if os.environ.get("BMX_BENCH_SELFTEST_RAISE") == str(rank):      # test hook: this rank's pipeline raises in the middle of the self-test
                raise RuntimeError("injected failure (BMX_BENCH_SELFTEST_RAISE)")
            bs = [to_dev(gen(b, rank), dev) for b in range(NBs)]
            torch.cuda.synchronize()                         # uploads run on torch's stream, the partition on the exchange stream
            tk = sg.route(Ds, *bs[0], exchange_now=True)
            for b in range(NBs):
                nxt = sg.route(Ds, *bs[b + 1]) if b + 1 < NBs else None
                sg.merge(tk)
                tk = nxt
            sg.ops.sync(); torch.cuda.synchronize()
            o = Oracle()
            o.load_rows(*sg.owned_resident_host(Rs, T0=T0, DT=DT))
            for b in range(NBs):
                for src in range(world):
                    cols = gen(b, src)
                    mine = synth.owner_of_np(cols[0], world) == rank
                    o.merge_batch(*[c[mine] for c in cols])
            ok = (not sg.overflowed()) and e.row_count() == len(o) and rows_digest(*e.dump_rows()) == o.digest()
            if not ok:
                why = "rank %d: the shard differs from the oracle after six batches through peer stores" % rank
            if os.environ.get("BMX_BENCH_SELFTEST_FAIL") == str(rank):     # test hook: pretend this rank saw wrong data
                ok, why = False, "injected mismatch (BMX_BENCH_SELFTEST_FAIL)"
        except Exception as err:
            ok, why = False, "rank %d: the direct exchange raised in the self-test: %s" % (rank, str(err)[:200])
        finally:
            if o is not None:
                o.close()
    okt = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev if dist.get_backend() != "gloo" else "cpu")
    dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    whys = [None] * world
    dist.all_gather_object(whys, why)
    for closer in ((lambda: sg.close()) if sg is not None else None, (lambda: sg.ops.close()) if sg is not None else None, (lambda: e.close()) if e is not None else None):
        if closer is not None:
            try:
                closer()
            except Exception as err:      # a context whose device-side wait expired carries a sticky error: it is being thrown away anyway
                print("bench[rank %d]: closing the self-test's context: %s" % (rank, str(err)[:160]), file=sys.stderr)
    reason = next((w for w in whys if w), None)
    if int(okt.item()) != 1:
        if rank == 0:
            print("bench: the direct exchange is refused on this machine (%s): the timed run uses the RCCL all-to-all" % reason, file=sys.stderr)
        return "rccl", reason
    return kind, (reason if kind != "direct" else None)


def cpu_baseline(n_batches=120, extras=True, n_batches_mt=24):
    """Oracle (C port of the reference merge rule) on ONE host core, same workload shape: ~10 s of timed CPU work (the contract's bounded sample), batches
    generated one at a time outside the timed intervals. extras: also the all-cores and the Node.js legs (N=1 line)."""
    from oracle.oracle import Oracle
    o = Oracle()
    o.load_rows(*gen_resident(R_PER_GPU))
    dt = 0.0
    for b in range(n_batches):
        batch = gen_batch(b, R_PER_GPU)
        t0 = time.perf_counter()
        o.merge_batch(*batch)
        dt += time.perf_counter() - t0
    o.close()
    out = {"value": n_batches * D_PER_STEP / dt, "unit": "merges/s", "cores": 1, "kind": "port",
           "sample": "%d x 1M-delta batches of the bench's own stream (config %d) against the 10M-row resident graph (load and batch generation excluded; %.1f s of timed CPU work), oracle/bmx_oracle.c, 1 thread" % (n_batches, CONFIG, dt)}
    if not extras:
        return out
    # extra line: the same port on all host cores, threads owning key shards
    try:
        from oracle.oracle import OracleMT
        T = max(1, min(os.cpu_count() or 1, 64))
        if hasattr(os, "sched_getaffinity"):
            T = max(1, min(T, len(os.sched_getaffinity(0))))
        m = OracleMT(T)
        m.load_rows(*gen_resident(R_PER_GPU))
        batches = [gen_batch(b, R_PER_GPU) for b in range(n_batches_mt)]
        t0 = time.perf_counter()
        for b in batches:
            m.merge_batch(*b)
        dtm = time.perf_counter() - t0
        m.close()
        del batches
        out["all_cores"] = {"value": n_batches_mt * D_PER_STEP / dtm, "unit": "merges/s", "cores": T, "kind": "port",
                            "sample": "the first %d of those batches; %d threads, thread k owns the keys with owner(id) == k and walks the whole batch" % (n_batches_mt, T)}
    except Exception as e:
        out["all_cores"] = {"error": str(e)[:200]}
    # the Node.js path on the same box: the per-delta processUpdate loop over a Map (the reference harness shape of BASELINE.md §2),
    # run by the golden-pinned JS twin of BulletCRT (bullet-js_amd/js/gpu-crt.js); bounded sample, one thread
    import shutil
    import subprocess
    node = shutil.which("node")
    if node:
        try:
            r = subprocess.run([node, os.path.join(ROOT, "bullet-js_amd", "js", "test", "cpu_baseline.js"), "1000000", "2000000"],
                               capture_output=True, text=True, timeout=240)
            j = json.loads(r.stdout.strip().splitlines()[-1])
            out["js_twin"] = {"value": j["value"], "unit": "merges/s", "cores": 1, "node": j["node"],
                              "sample": "%d deltas (10 %% inserts) against %d resident keys, processUpdate loop over a Map, 1 thread" % (j["deltas"], j["resident_keys"])}
        except Exception as e:  # the JS figure is an extra; never fail the bench for it
            out["js_twin"] = {"error": str(e)[:200]}
    return out


def numa_node_of_cpu(cpu, sysfs="/sys/devices/system/node", n_cpus=None):
    """(node name, its cpulist string) of the NUMA node that holds `cpu`, or None when the topology cannot be read or has one node only."""
    import glob
    n_cpus = (os.cpu_count() or 0) if n_cpus is None else n_cpus
    for d in sorted(glob.glob(os.path.join(sysfs, "node[0-9]*"))):
        cl = open(os.path.join(d, "cpulist")).read().strip()
        ids = set()
        for part in cl.split(","):
            lo, _, hi = part.partition("-")
            ids.update(range(int(lo), int(hi or lo) + 1))
        if cpu in ids:
            return (os.path.basename(d), cl) if len(ids) < n_cpus else None      # (one node = the whole machine: nothing to pin)
    return None


def js_host_rate():
    """End-to-end rate of the JS host on this box (extra key, N=1 only): sync-chunk entries -> path hashing -> typed columns -> N-API ->
    GPU merge -> winners, one thread; and the same with the keys already hashed. bullet-js_amd/js/test/e2e_rate.js, bounded sample."""
    import shutil
    import subprocess
    node = shutil.which("node")
    if not node:
        return None
    # The host-bound figures depend on where the one node thread and its memory sit: on a two-socket box the scheduler moves it between the sockets and
    # the store-kept seam swings between 0.46 and 0.66 M entries/s; kept on the cores of ONE NUMA node (the one this process runs on) it stays at
    # 0.62-0.69 (profiles/r04_e2e_apply.log). taskset only, and only when the topology can be read; said in the line (`cpus`).
    pin, cpus = [], None
    try:
        here = int(open("/proc/self/stat").read().rsplit(")", 1)[1].split()[36])      # field 39: the CPU this thread last ran on
        found = numa_node_of_cpu(here) if shutil.which("taskset") else None
        if found:
            pin, cpus = ["taskset", "-c", found[1]], "%s: CPUs %s" % found
    except Exception:
        pin, cpus = [], None
    try:
        script = os.path.join(ROOT, "bullet-js_amd", "js", "test", "e2e_rate.js")
        r = subprocess.run(pin + [node, script, "1000000", "500000", "8"], capture_output=True, text=True, timeout=240)
        j = json.loads(r.stdout.strip().splitlines()[-1])
        for key, out_key in (("apply", "applied_path"), ("lazy", "lazy_path"), ("vector", "vector")):      # one process each: a 4M-entry run of every section at once does not fit node 12's default heap
            try:
                r2 = subprocess.run(pin + [node, script, "1000000", "200000", "5", "only", key], capture_output=True, text=True, timeout=300)
                j2 = json.loads(r2.stdout.strip().splitlines()[-1])
                j[out_key] = j2[out_key]
            except Exception as e:
                j[out_key] = {"error": str(e)[:200]}
        j["cpus"] = cpus or "not pinned"
        j["applied_path_sample"] = "the real ingestion seam with the STORE KEPT: attach(bullet, {batchSync}) -> processSyncEntries over 5 chunks of 200k sync entries against 1M resident nodes (90 % updates, 10 % new nodes): every winner replaces its node in the nested store, meta[path] gets its clock, the op log and the put queue are fed (src/bullet.js:184-266 per batch); beside it the same entries one by one through setData and the host resolver (the reference's loop body)"
        j["vector_sample"] = "5 chunks of 200k entries under clocks over ordered subsets of three writers (N4): nodes' clock rows in the device's vector-clock table, synchronous GpuCRT.mergeEntries"
        j["sample"] = "8 chunks of 500k sync entries (10 % new nodes) against 1M resident nodes per figure, node-level resolution (one clock-row delta per entry + the winners' value rows); GpuCRT.mergeEntries (synchronous), GpuCRT.mergeEntriesPipelined (mergeEntriesAsync, two chunks in flight) and GpuCRT.mergeBatch (typed columns) over the N-API addon, page-locked host columns"
        return j
    except Exception as e:
        return {"error": str(e)[:200]}


def verify_against_oracle(eng, host_batches, winners_dev, n_applied_dev, resident_cols):
    """Replay the very batches the device merged (warm-up + timed) through the CPU oracle and compare every batch's winner list,
    the row count and the digest of the whole table. Runs after the timed region; raises SystemExit on any difference."""
    from oracle.oracle import Oracle, rows_digest
    t0 = time.perf_counter()
    o = Oracle()
    o.load_rows(*resident_cols)
    n_app = n_applied_dev.cpu().numpy()
    checked = 0
    for b, cols in enumerate(host_batches):
        _, want = o.merge_batch(*cols)
        got = winners_dev[b][: int(n_app[b])].cpu().numpy().view(np.uint32)
        if len(got) != len(want) or not np.array_equal(got, want):
            raise SystemExit("VERIFICATION FAILED: batch %d: device reports %d winners, oracle %d (or different indices)" % (b, len(got), len(want)))
        checked += len(want)
    rows = eng.row_count()
    if rows != len(o):
        raise SystemExit("VERIFICATION FAILED: %d rows on the device, %d in the oracle" % (rows, len(o)))
    dg = rows_digest(*eng.dump_rows())
    if dg != o.digest():
        raise SystemExit("VERIFICATION FAILED: table digest %x != oracle %x" % (dg, o.digest()))
    o.close()
    return {"against": "oracle/bmx_oracle.c replaying the same batches after the timed region", "batches": len(host_batches), "winner_indices_compared": checked,
            "rows": rows, "table_digest": "%016x" % dg, "ok": True, "seconds": round(time.perf_counter() - t0, 2)}


def scan_bench(bmx, dev, R, reps=20, wide=False):
    """Config 3: range/equals scans over an indexed field of R nodes: int32 values in [0,1000) or, with `wide`, the same values shifted
    beyond 32 bits (the int64 column variant of SURVEY 8(d) config 3). Whole-scan time = HIP events around `reps` back-to-back scans;
    the mask kernel's own time comes from per-kernel HIP events (bmx_profile_read_scan) in a second pass.
    Algorithmic bytes: whole scan w*R (value column, w = 4 or 8) + 8*M (ids out); mask kernel w*R."""
    from bmx import synth
    fa = synth.fnv1a32("n:score" if wide else "n:age")
    w = 8.0 if wide else 4.0
    sh = 33 if wide else 0
    out = {"rows": R, "column": "int64" if wide else "int32"}
    tscan = {}
    try:      # HBM bytes per launch from the committed rocprofv3 --pmc passes over this same scan (profiles/r02_traffic_scan_passes.sh), 100M rows only
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_scan.json")))
        if tj.get("rows") == R:
            tscan = {q.split("/", 1)[1]: e for q, e in tj["queries"].items() if q.startswith(out["column"] + "/")}
    except Exception:
        tscan = {}
    QUERIES = [("equals_0.1pct", 42, 42), ("range_1pct", 100, 109), ("range_10pct", 100, 199), ("range_50pct", 0, 499)]
    want = {name: [0, 0] for name, _, _ in QUERIES}     # per query: match count and the wrap-around sum of the matching ids, from numpy while the rows are generated
    with bmx.Engine(capacity_rows=R + 1024 + 4 * D_PER_STEP, device=dev.index or 0) as e:
        for r0 in range(0, R, 10_000_000):          # load in 10M-row pieces: bounded host memory
            m = min(10_000_000, R - r0)
            ids = synth.splitmix64_np(np.arange(r0 + 1, r0 + m + 1, dtype=np.uint64))
            with np.errstate(over="ignore"):
                ages = (synth.splitmix64_np(ids ^ np.uint64(0xABCDEF)) % np.uint64(1000)).astype(np.int64)
                for name, lo, hi in QUERIES:
                    sel = (ages >= lo) & (ages <= hi)
                    want[name][0] += int(sel.sum()); want[name][1] = (want[name][1] + int(ids[sel].sum(dtype=np.uint64))) & ((1 << 64) - 1)
            e.load_rows(ids, np.full(m, fa, np.uint32), np.full(m, 5, np.int64), ages << sh)
        del ids, ages
        t0 = time.perf_counter(); e.index_build(fa); out["index_first_build_ms"] = round((time.perf_counter() - t0) * 1e3, 3)   # + one-time allocation of the maintenance map
        e.index_drop(fa)
        t0 = time.perf_counter(); e.index_build(fa); out["index_build_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        out_ids = torch.zeros(R, dtype=torch.int64, device=dev)
        out_pos = torch.zeros(R, dtype=torch.int32, device=dev)
        id_col = torch.zeros(R, dtype=torch.int64, device=dev)
        n_out = torch.zeros(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()                    # torch zero-fills on ITS stream: the fills must have run before the engine writes these buffers on its own
        e.index_ids_dev(fa, 0, R, id_col)           # the index's id column: what a position stands for (verification of the position output)
        checked = 0

        def i64(x):
            return x - (1 << 64) if x >= (1 << 63) else x

        for name, lo, hi in QUERIES:
            for _ in range(3):
                e.scan_range_dev(fa, lo << sh, hi << sh, out_ids, R, n_out)
            e.sync(); e.timer_start()
            for _ in range(reps):
                e.scan_range_dev(fa, lo << sh, hi << sh, out_ids, R, n_out)
            ms = e.timer_stop() / reps
            m = int(n_out.item())
            # every timed scan answers the same query: count and id checksum against numpy (after the timed region)
            if m != want[name][0] or int(out_ids[:m].sum().item()) != i64(want[name][1]):
                raise SystemExit("VERIFICATION FAILED: scan %s over %d %s rows: %d matches (numpy: %d) or a different id checksum" % (name, R, out["column"], m, want[name][0]))
            # the same query with POSITION output (u32 index positions, no id gather): bmx_scan_range_pos
            for _ in range(3):
                e.scan_range_pos_dev(fa, lo << sh, hi << sh, out_pos, R, n_out)
            e.sync(); e.timer_start()
            for _ in range(reps):
                e.scan_range_pos_dev(fa, lo << sh, hi << sh, out_pos, R, n_out)
            ms_pos = e.timer_stop() / reps
            mp = int(n_out.item())
            got_sum = int(id_col[out_pos[:mp].long()].sum().item()) if mp == m else None
            if mp != m or got_sum != i64(want[name][1]):
                raise SystemExit("VERIFICATION FAILED: position scan %s over %d %s rows names other rows than the id scan (%d positions for %d ids; id checksum through the positions %s, expected %d)" %
                                 (name, R, out["column"], mp, m, got_sum, i64(want[name][1])))
            checked += 2
            # the mask pass alone, un-bracketed: the count-only form of the same query (k_scan_mask without the mask write + a one-workgroup sum), back to back
            e.sync(); e.timer_start()
            for _ in range(reps):
                e.scan_range_dev(fa, lo << sh, hi << sh, None, 0, n_out)
            ms_cnt = e.timer_stop() / reps
            e.profile_enable(True)
            for _ in range(8):
                e.scan_range_dev(fa, lo << sh, hi << sh, out_ids, R, n_out)
            kms, _ = e.profile_read_scan()
            e.profile_enable(False)
            alg = w * R + 8.0 * m
            alg_pos = w * R + 4.0 * m
            mask_s = ms_cnt * 1e-3
            out[name] = {"matches": m, "us": round(ms * 1e3, 2), "achieved_GBs": round(alg / (ms * 1e-3) / 1e9, 1), "frac_of_8TBs": round(alg / (ms * 1e-3) / 8e12, 4),
                         "rows_per_s": round(R / (ms * 1e-3)),
                         "position_output": {"us": round(ms_pos * 1e3, 2), "achieved_GBs": round(alg_pos / (ms_pos * 1e-3) / 1e9, 1), "frac_of_8TBs": round(alg_pos / (ms_pos * 1e-3) / 8e12, 4),
                                             "algorithmic_bytes": "w*R + 4*M"},
                         "roofline_mask_kernel": {"bound": "hbm", "kernel": "k_scan_mask", "achieved": round(w * R / mask_s / 1e9, 1) if mask_s > 0 else None, "peak": HBM_PEAK_GBS,
                                                  "unit": "GB/s", "frac": round(w * R / mask_s / 1e9 / HBM_PEAK_GBS, 4) if mask_s > 0 else None,
                                                  "timed_as": "count-only scan (k_scan_mask without the mask write + one-workgroup sum), back to back, no event brackets: an upper bound of the kernel's duration",
                                                  "count_only_scan_us": round(ms_cnt * 1e3, 2),
                                                  "traffic": tscan.get(name, {}).get("mask", {}).get("bytes_per_launch"),
                                                  "traffic_emit": tscan.get(name, {}).get("emit", {}).get("bytes_per_launch"),
                                                  "kernel_us_between_events": {"scan_mask": round(kms["scan_mask"] * 1e3, 2), "offsets_and_emit": round(kms["emit"] * 1e3, 2)}}}
        out["verified"] = {"against": "numpy over the generated rows: match count and wrap-around sum of the matching ids, id output and position output (ids gathered through the index's id column)",
                           "scans_checked": checked, "ok": True}
        # The same queries through the VALUE-ORDERED VIEW of the index (bmx_index_set_ordered: the reference's index is a Map keyed by value,
        # src/bullet-query.js:30-73): two k-ary searches + one contiguous copy. Bytes moved are 16 / 8 per match (ids / positions), not the column.
        ov = {}
        e.index_set_ordered(fa, 1)
        e.sync(); t0 = time.perf_counter()
        e.scan_range_dev(fa, 42 << sh, 42 << sh, None, 0, n_out); e.sync()          # the first query sorts the view
        ov["sort_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        if not e.index_ordered_info(fa)[1]:
            raise SystemExit("the value-ordered view of the %d-row %s index was not built" % (R, out["column"]))
        for name, lo, hi in QUERIES:
            for _ in range(3):
                e.scan_range_dev(fa, lo << sh, hi << sh, out_ids, R, n_out)
            e.sync(); e.timer_start()
            for _ in range(reps):
                e.scan_range_dev(fa, lo << sh, hi << sh, out_ids, R, n_out)
            ms = e.timer_stop() / reps
            m = int(n_out.item())
            if m != want[name][0] or int(out_ids[:m].sum().item()) != i64(want[name][1]):
                raise SystemExit("VERIFICATION FAILED: ordered-view scan %s over %d %s rows: %d matches (numpy: %d) or a different id checksum" % (name, R, out["column"], m, want[name][0]))
            for _ in range(3):
                e.scan_range_pos_dev(fa, lo << sh, hi << sh, out_pos, R, n_out)
            e.sync(); e.timer_start()
            for _ in range(reps):
                e.scan_range_pos_dev(fa, lo << sh, hi << sh, out_pos, R, n_out)
            ms_pos = e.timer_stop() / reps
            mp = int(n_out.item())
            got_sum = int(id_col[out_pos[:mp].long()].sum().item()) if mp == m else None
            if mp != m or got_sum != i64(want[name][1]):
                raise SystemExit("VERIFICATION FAILED: ordered-view position scan %s over %d %s rows names other rows than numpy" % (name, R, out["column"]))
            e.sync(); e.timer_start()
            for _ in range(reps):
                e.scan_range_dev(fa, lo << sh, hi << sh, None, 0, n_out)
            ms_cnt = e.timer_stop() / reps
            checked += 2
            ov[name] = {"matches": m, "us": round(ms * 1e3, 2), "bytes_moved": 16 * m, "moved_GBs": round(16.0 * m / (ms * 1e-3) / 1e9, 1),
                        "position_output_us": round(ms_pos * 1e3, 2), "count_only_us": round(ms_cnt * 1e3, 2),
                        "speedup_over_the_column_scan": round(out[name]["us"] / (ms * 1e3), 2)}
        ov["note"] = ("opt-in per index; kept current under writes by patching (round 5); matches come in (value, position) order; every timed query verified against numpy like the scans above")
        checked_here = checked

        def delta_batch(seed, first_new):
            """1M deltas on the indexed field: 90 % updates of existing nodes, 10 % new nodes, every one under a clock above everything stored"""
            rng = np.random.default_rng(seed)
            rows = rng.integers(0, R, D_PER_STEP).astype(np.int64)
            rows[::10] = first_new + np.arange(len(rows[::10]))
            bid = synth.splitmix64_np((rows + 1).astype(np.uint64))
            with np.errstate(over="ignore"):
                bval = ((synth.splitmix64_np(bid ^ np.uint64(0x5151 + seed)) % np.uint64(1000)).astype(np.int64)) << sh
            return (torch.from_numpy(bid.view(np.int64)).to(dev), torch.full((D_PER_STEP,), int(np.array([fa], np.uint32).view(np.int32)[0]), dtype=torch.int32, device=dev),
                    torch.full((D_PER_STEP,), 9 + seed, dtype=torch.int64, device=dev), torch.from_numpy(bval).to(dev))

        # the view UNDER WRITES (the reference moves a path between value buckets on every write, src/bullet-query.js:139-176): a 1M-delta merge on
        # the indexed field, then the first equals (refresh of the columns from the change log + sort of the change run + one streaming merge into the view), then more
        # (a full period of the scheme first, untimed — at 10^8 rows four cycles, the last of which makes the first rewrite of main due: the first patch of an index allocates
        # the view's second set of columns, the pending patch and the sort scratch, the first host-mode answer its download buffer, and every kernel's first launch loads its
        # code — all once in the index's or the process's life. What is timed below is the steady state.)
        hb = bmx.HostBuffer((1 << 18) * 8); host_ids = hb.array(np.uint64, 1 << 18); host_ids[:] = 0      # the caller's answer buffer (page-locked: what a host that cares uses)
        WARM = 4
        for wc in range(WARM):
            cols = delta_batch(6 + wc, R + wc * D_PER_STEP)
            torch.cuda.synchronize()
            e.merge_batch_dev(D_PER_STEP, *cols, bmx.INSERT_REFERENCE, applied=None, n_applied=n_out)
            e.scan_range(fa, 42 << sh, 42 << sh, out=host_ids); e.sync()
        # FOUR timed cycles: the pending patch grows over them (empty -> ~6 % of the rows) and the last makes a rewrite of main due, so every state of the scheme is timed.
        # Each is a HOST-mode equals, ids in host memory when it returns: what the JS host calls. The refresh, the patch and the answer are in front of the return; a rewrite
        # of the view's main run that the patch made due is enqueued BEHIND the answer and not waited for (a device-mode call followed by bmx_sync would wait for it).
        cycles = []
        s_first = e.index_ordered_stats(fa)
        for cyc in range(4):
            cols = delta_batch(6 + WARM + cyc, R + (WARM + cyc) * D_PER_STEP)
            torch.cuda.synchronize()
            s0 = e.index_ordered_stats(fa)
            e.merge_batch_dev(D_PER_STEP, *cols, bmx.INSERT_REFERENCE, applied=None, n_applied=n_out)
            e.sync()
            t0 = time.perf_counter()
            got_ids = e.scan_range(fa, 42 << sh, 42 << sh, out=host_ids)
            first_us = (time.perf_counter() - t0) * 1e6
            m_view = len(got_ids)
            with np.errstate(over="ignore"):
                sum_view = int(got_ids.view(np.int64).sum(dtype=np.int64))
            t0 = time.perf_counter(); e.sync()
            behind_us = (time.perf_counter() - t0) * 1e6       # (a rewrite of main running behind the answer, when this patch made one due)
            s1 = e.index_ordered_stats(fa)
            cycles.append({"first_equals_us": round(first_us, 1), "work_behind_the_answer_us": round(behind_us, 1), "patch_us": round(s1["last_patch_us"], 1),
                           "patch_keys": s1["keys_patched"] - s0["keys_patched"], "pending_keys_after": s1["pending_keys"], "matches": m_view})
        s1 = e.index_ordered_stats(fa)
        ov["cycles"] = cycles
        firsts = sorted(c["first_equals_us"] for c in cycles)
        ov["first_equals_after_a_1M_delta_merge_us"] = firsts[-1]                     # the WORST of the four states (the figure the line carries)
        ov["first_equals_after_a_1M_delta_merge_us_best"] = firsts[0]
        ov["work_behind_that_answer_us"] = max(c["work_behind_the_answer_us"] for c in cycles)
        ov["main_rewrites_in_the_cycles"] = s1["rewrites"] - s_first["rewrites"]
        ov["view_kept_current_by"] = ("patch" if s1["patches"] - s_first["patches"] == 4 and s1["sorts"] == s_first["sorts"] and e.index_ordered_info(fa)[1]
                                      else ("sort" if s1["sorts"] > s_first["sorts"] else "column scan (view stale)"))
        ov["patch_us"] = cycles[-1]["patch_us"]; ov["patch_keys"] = cycles[-1]["patch_keys"]; ov["pending_keys_after"] = s1["pending_keys"]
        e.sync(); e.timer_start()
        for _ in range(reps):
            e.scan_range_dev(fa, 42 << sh, 42 << sh, out_ids, R, n_out)
        ov["next_equals_us"] = round(e.timer_stop() / reps * 1e3, 2)
        e.index_set_ordered(fa, 0)                  # off: the same query by the column scan, on the same (already refreshed) columns, must name the same rows
        e.scan_range_dev(fa, 42 << sh, 42 << sh, out_ids, R, n_out); e.sync()
        m_scan = int(n_out.item())
        if m_scan != m_view or int(out_ids[:m_scan].sum().item()) != sum_view or ov["view_kept_current_by"] != "patch":
            raise SystemExit("VERIFICATION FAILED: after a 1M-delta merge the value-ordered view of the %d-row %s index answered equals with %d rows, the column scan with %d (or another id checksum); kept current by: %s" %
                             (R, out["column"], m_view, m_scan, ov["view_kept_current_by"]))
        checked += 1
        out["ordered_view"] = ov
        out["verified"]["scans_checked"] = checked
        del out_pos, id_col
        # index maintenance WITHOUT a view: another 1M-delta merge on the indexed field (90 % updates of existing nodes, 10 % new nodes), then the first scan — which brings
        # the index up to date from the merge's change log instead of rebuilding it from the table (include/bmx.h "Maintenance")
        cols = delta_batch(6 + WARM + 4, R + (WARM + 4) * D_PER_STEP)
        full0, inc0 = e.index_refresh_counts()
        torch.cuda.synchronize()                    # the batch columns were produced on torch's stream
        e.merge_batch_dev(D_PER_STEP, *cols, bmx.INSERT_REFERENCE, applied=None, n_applied=n_out)
        e.sync()
        t0 = time.perf_counter()
        e.scan_range_dev(fa, 42 << sh, 42 << sh, out_ids, R, n_out); e.sync()
        first_us = (time.perf_counter() - t0) * 1e6
        full1, inc1 = e.index_refresh_counts()
        out["first_scan_after_a_1M_delta_merge"] = {"us": round(first_us, 1), "index_brought_up_to_date_by": "change log" if inc1 > inc0 and full1 == full0 else "rebuild",
                                                    "matches": int(n_out.item()), "index_rows": e.index_size(fa)}
    return out


LINE_LIMIT = 4000      # bytes: the driver keeps an ~8 KB tail of stdout; the line stays under half of that whatever sections a run adds


def _r(x, nd=4):
    return round(x, nd) if isinstance(x, float) else x


def compact_line(out, detail_path=None):
    """The ONE stdout line (round 4's 21 KB line was cut by the driver and parsed as nothing). Contract fields first, then
    `roofline`, `cpu_baseline`, `verified`, then one-number summaries of the optional sections; tables, prose and per-rank digests live in
    `bench_detail.json` (named in the line) and on stderr. Sections are dropped from the end, never the contract fields, should a run ever exceed LINE_LIMIT."""
    g = out.get
    line = {k: g(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["value"] = _r(line["value"], 1); line["ms_per_step"] = _r(line["ms_per_step"], 6)
    cfg = dict(g("config") or {})
    cfg["workload"] = str(cfg.get("workload", ""))[:260]
    line["config"] = cfg
    rf = g("roofline") or {}
    line["roofline"] = {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "kernel_ms",
                                                "launches_averaged", "requests_per_launch", "whole_merge_achieved_GBs", "traffic_source") if k in rf}
    if isinstance(line["roofline"].get("requests_per_launch"), dict):
        line["roofline"]["requests_per_launch"] = line["roofline"]["requests_per_launch"].get("total")
    if "traffic_source" in line["roofline"]:
        line["roofline"]["traffic_source"] = "N=1 PMC passes (profiles/traffic_probe_apply.json)"
    cb = g("cpu_baseline")
    if isinstance(cb, dict):
        c = {k: _r(cb.get(k), 1) for k in ("value", "unit", "cores", "kind")}
        c["sample"] = str(cb.get("sample", ""))[:200]
        for leg in ("all_cores", "js_twin"):
            if isinstance(cb.get(leg), dict) and "value" in cb[leg]:
                c[leg] = {"value": _r(cb[leg]["value"], 1), "cores": cb[leg].get("cores")}
        line["cpu_baseline"] = c
    else:
        line["cpu_baseline"] = None
    v = g("verified")
    if isinstance(v, dict):
        line["verified"] = {k: v[k] for k in ("ok", "batches", "winner_indices_compared", "rows", "table_digest") if k in v}
        if "per_rank" in v:
            line["verified"]["ranks_ok"] = sum(1 for x in v["per_rank"] if x.get("ok"))
    else:
        line["verified"] = None
    optional = []      # (key, value) in the order they are given up if the line grows too long: last first
    for k in ("event_ms_per_step", "winners_per_step", "host_enqueue_ms_per_step"):
        if g(k) is not None:
            optional.append((k, g(k)))
    ex = g("exchange")
    if isinstance(ex, dict):
        optional.append(("exchange", {k: (str(ex[k])[:160] if isinstance(ex[k], str) else ex[k]) for k in ("kind", "refused", "why", "records_sent_to_other_shards", "bytes_per_record", "steps") if k in ex}))
    sc = g("scan_config3")
    if isinstance(sc, dict):
        s3 = {}
        for size, e in sc.items():
            if not isinstance(e, dict):
                continue
            one = {}
            q10 = e.get("range_10pct") or {}
            fr = sorted(f for f in ((e.get(q) or {}).get("roofline_mask_kernel", {}).get("frac") for q in ("equals_0.1pct", "range_1pct", "range_10pct", "range_50pct")) if f is not None)
            if fr:
                one["mask_frac"] = fr[len(fr) // 2] if len(fr) % 2 else round((fr[len(fr) // 2 - 1] + fr[len(fr) // 2]) / 2, 4)      # median over the four queries
            if q10:
                one["range10_ids_us"] = q10.get("us")
                one["range10_pos_us"] = (q10.get("position_output") or {}).get("us")
            q50 = e.get("range_50pct") or {}
            if q50:
                one["range50_ids_us"] = q50.get("us")
            eq = e.get("equals_0.1pct") or {}
            if eq:
                one["equals_us"] = eq.get("us")
            ov = e.get("ordered_view") or {}
            if ov:
                one["view_equals_us"] = (ov.get("equals_0.1pct") or {}).get("us")
                one["view_range10_ids_us"] = (ov.get("range_10pct") or {}).get("us")
                one["view_sort_ms"] = ov.get("sort_ms")
                for k, short in (("first_equals_after_a_1M_delta_merge_us", "view_first_equals_after_merge_us_worst_of_4"), ("first_equals_after_a_1M_delta_merge_us_best", "view_first_equals_after_merge_us_best_of_4"), ("next_equals_us", "view_next_equals_us"), ("view_kept_current_by", "view_kept_current_by")):
                    if k in ov:
                        one[short] = ov[k]
            fs = e.get("first_scan_after_a_1M_delta_merge") or {}
            if fs:
                one["first_scan_after_merge_us"] = fs.get("us")
            one["ok"] = bool((e.get("verified") or {}).get("ok"))
            s3[size] = one
        optional.append(("scan_config3", s3))
    jh = g("js_host")
    if isinstance(jh, dict):
        ap, vc, lz = jh.get("applied_path") or {}, jh.get("vector") or {}, jh.get("lazy_path") or {}
        j = {"mergeEntries_per_s": jh.get("mergeEntries_per_s"), "mergeEntriesPipelined_per_s": jh.get("mergeEntriesPipelined_per_s"),
             "store_kept_entries_per_s": ap.get("batchSync_apply_entries_per_s"), "store_kept_lazy_entries_per_s": lz.get("batchSync_lazy_entries_per_s"),
             "store_kept_lazy_incl_fold_entries_per_s": lz.get("incl_the_fold_entries_per_s"), "vector_pipelined_per_s": vc.get("mergeEntriesPipelined_per_s")}
        if "error" in jh:
            j = {"error": str(jh["error"])[:120]}
        optional.append(("js_host", {k: _r(x, 0) for k, x in j.items()}))
    um = g("unique_keys_mode")
    if isinstance(um, dict):
        optional.append(("unique_keys_probe_ms", (um.get("kernel_ms") or {}).get("probe_apply")))
    tp = g("table_placement")
    if isinstance(tp, dict):
        optional.append(("table_placement", {k: tp[k] for k in ("candidates", "probe_us_chosen", "probe_us_slowest", "policy") if k in tp}))
    if detail_path:
        line["detail"] = os.path.basename(detail_path)
    for k, val in optional:
        line[k] = val
    text = json.dumps(line, separators=(", ", ": "))
    drop = [k for k, _ in optional][::-1]
    while len(text) >= LINE_LIMIT and drop:
        line.pop(drop.pop(0), None)
        text = json.dumps(line, separators=(", ", ": "))
    if len(text) >= LINE_LIMIT:          # cannot happen with the bounded fields above; never print a line the driver would cut
        line["config"] = {"workload": cfg["workload"][:120]}
        text = json.dumps(line, separators=(", ", ": "))
    return text


def emit(out, real_stdout, detail_path=None):
    """full record -> bench_detail.json (+ stderr), short line -> stdout"""
    detail_path = detail_path or os.environ.get("BMX_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
    full = json.dumps(out)
    try:
        with open(detail_path, "w") as f:
            f.write(full + "\n")
    except OSError as e:
        print("bench: could not write %s: %s" % (detail_path, e), file=sys.stderr)
        detail_path = None
    print("bench detail: " + full, file=sys.stderr)
    real_stdout.write(compact_line(out, detail_path) + "\n")
    real_stdout.flush()


WAIT_LIMIT_S = 20.0          # device-side waits of the exchange (bmx_set_wait_limit): a peer 20 s late for a 100-us step is gone
COLL_TIMEOUT_S = 150.0       # per collective / barrier / rendezvous inside the ranks
RANK_TIMEOUT_S = 420.0      # --rank-timeout: wall-clock limit of the launcher on its child (the N ranks), first contact with 8 GPUs included


def _tail_lines(path, n=12):
    try:
        with open(path, "rb") as f:
            f.seek(0, 2)
            f.seek(max(0, f.tell() - 16384))
            return f.read().decode("utf-8", "replace").splitlines()[-n:]
    except OSError:
        return []


def launch_ranks(n_gpus, argv, timeout_s=None):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process is only the LAUNCHER. It starts
    `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD process (its own process group), hands rank 0's single
    JSON line through and exits with the child's return code. It never touches the GPU (no HIP call, no torch.cuda.is_available(): a process that
    has initialised the GPU must neither exec nor be needed for anything here), it does not retry, and a failing child fails the run.
    BOUNDED: after `timeout_s` seconds of wall clock the child's whole process group is terminated (SIGTERM, SIGKILL 10 s later),
    the last stderr lines of every rank are printed (torch.distributed.run tees each rank's stderr into a log directory) and the launcher exits
    with 124 — a hung rendezvous, IPC open or barrier then costs the limit, not the driver's whole slot. A result line that has ALREADY arrived is
    still delivered if the ranks only hang while shutting down. -> exit code"""
    import shutil
    import signal
    import socket
    import subprocess
    import tempfile
    import threading
    timeout_s = RANK_TIMEOUT_S if timeout_s is None else float(timeout_s)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:      # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")                  # dmabuf IPC: what RCCL and the direct exchange's mappings need on this driver
    env["BMX_BENCH_LAUNCHED_AT"] = repr(time.time())                   # the ranks stamp their progress lines relative to this
    logdir = tempfile.mkdtemp(prefix="bmx_bench_ranks_")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "--log-dir", logdir, "--tee", "2", os.path.abspath(__file__)] + list(argv)
    print("bench: launching %d ranks (limit %.0f s): %s" % (n_gpus, timeout_s, " ".join(cmd)), file=sys.stderr)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, start_new_session=True)
    got = {"line": None}

    def pump():
        for ln in child.stdout:                                        # rank 0 prints exactly one JSON line; anything else on stdout goes to stderr
            t = ln.strip()
            if t.startswith("{") and t.endswith("}") and '"metric"' in t:
                got["line"] = t
            elif t:
                print(t, file=sys.stderr)
    th = threading.Thread(target=pump, daemon=True)
    th.start()

    def forward(signum, frame):                                        # the ranks live in their own session: a signal to the launcher must reach them too
        try:
            os.killpg(child.pid, signal.SIGTERM)
        except Exception:
            pass
        raise SystemExit(128 + signum)
    for sg_ in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        try:
            signal.signal(sg_, forward)
        except (ValueError, OSError):                                  # not the main thread (tests): nothing to forward
            pass

    def stop_group():
        for sig, grace in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 5.0)):
            try:
                os.killpg(child.pid, sig)                              # exactly the process group this launcher started
            except (ProcessLookupError, PermissionError, AttributeError, OSError):
                try:
                    child.kill()
                except Exception:
                    pass
            try:
                child.wait(timeout=grace)
                return
            except subprocess.TimeoutExpired:
                continue

    def rank_tails():
        import glob
        logs = sorted(glob.glob(os.path.join(logdir, "**", "stderr.log"), recursive=True))
        for lg in logs:
            print("bench: last stderr lines of %s" % os.path.relpath(lg, logdir), file=sys.stderr)
            for ln in _tail_lines(lg):
                print("    " + ln, file=sys.stderr)
        if not logs:
            print("bench: no per-rank logs under %s" % logdir, file=sys.stderr)

    deadline = time.monotonic() + timeout_s
    rc = None
    timed_out = False
    while rc is None:
        try:
            rc = child.wait(timeout=max(0.05, min(1.0, deadline - time.monotonic())))
        except subprocess.TimeoutExpired:
            if time.monotonic() >= deadline:
                timed_out = True
                break
            if got["line"] is not None and deadline - time.monotonic() > 45.0:
                deadline = time.monotonic() + 45.0                     # the result is in: the ranks get 45 s to shut down, not the rest of the limit
    if timed_out:
        print("bench: the ranks did not finish within %.0f s: terminating process group %d" % (timeout_s, child.pid), file=sys.stderr)
        stop_group()
        th.join(timeout=5.0)
        rank_tails()
        shutil.rmtree(logdir, ignore_errors=True)
        if got["line"] is not None:                                   # measured, verified and printed; only the shutdown hung
            print("bench: the result line had arrived before the limit: delivering it", file=sys.stderr)
            sys.stdout.write(got["line"] + "\n"); sys.stdout.flush()
            return 0
        return 124
    th.join(timeout=10.0)
    if rc != 0:
        print("bench: the ranks exited with code %d: no result" % rc, file=sys.stderr)
        rank_tails()
        shutil.rmtree(logdir, ignore_errors=True)
        return rc
    shutil.rmtree(logdir, ignore_errors=True)
    if got["line"] is None:
        print("bench: the ranks finished without printing a result line", file=sys.stderr)
        return 1
    sys.stdout.write(got["line"] + "\n")
    sys.stdout.flush()
    return 0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 20 for config 2, 90 for config 5)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default: 3 for config 2, 10 for config 5)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 5], help="BASELINE.json configs[1]/[3] (2) or configs[4], streaming hot-key replay (5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-run comparison with the CPU oracle")
    ap.add_argument("--no-scan", action="store_true", help="skip the config-3 index scans (10M and 100M rows)")
    ap.add_argument("--scan", action="store_true", help="(kept for compatibility: the scans are on by default)")
    ap.add_argument("--scan-rows", type=str, default="10000000,100000000", help="comma-separated index sizes of the config-3 scans")
    ap.add_argument("--force-sharded", action="store_true", help="run the N>1 code path (partition + all-to-all + merge) even with one rank: rehearsal only")
    ap.add_argument("--rank-timeout", type=float, default=RANK_TIMEOUT_S, help="launcher role (--gpus N > 1 without WORLD_SIZE): seconds after which the ranks' process group is terminated and the run fails")
    ap.add_argument("--no-defer", action="store_true", help="A/B switch: keep every batch's winner compaction on the merge stream (bmx_set_deferred_compaction(0))")
    return ap.parse_args(argv)


def main(argv=None):
    global CONFIG
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launcher role: nothing below this line runs in this process (in particular no GPU call)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:] if argv is None else argv, timeout_s=args.rank_timeout))
    # stdout carries exactly ONE JSON line: native libraries (RCCL's version banner) print to fd 1, so fd 1 is pointed
    # at stderr for the whole run and the JSON goes to a private duplicate of the real stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    CONFIG = args.config
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # Config 2 keeps the 20 + 3 steps of rounds 1-2 so that the rounds compare: the stream gets EASIER as it goes on (every batch raises the clocks of the
    # rows it hits, so later batches win less often: 838k winners per step over steps 3..22, 784k over steps 10..69), which a longer default would book as speed.
    K = args.steps if args.steps is not None else (90 if CONFIG == 5 else 20)
    W = args.warmup if args.warmup is not None else (10 if CONFIG == 5 else 3)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # BMX_BENCH_ONE_GPU_REHEARSAL=1: all ranks share cuda:0 and talk over gloo — exercises this file's N>1 logic on a one-GPU box
    # (RCCL refuses two ranks on one device); its numbers mean nothing
    rehearsal = os.environ.get("BMX_BENCH_ONE_GPU_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import bmx
    dist = None
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
        # every rendezvous, collective and barrier of the run is time-limited: a rank that never arrives fails the others after COLL_TIMEOUT_S
        # (the process group's watchdog aborts the process), well inside the launcher's limit, instead of holding them for ever
        import datetime
        tmo = datetime.timedelta(seconds=float(os.environ.get("BMX_BENCH_COLL_TIMEOUT", COLL_TIMEOUT_S)))
        say("joining the process group (%s, world %d, collective timeout %.0f s)" % ("gloo rehearsal" if rehearsal else "nccl", world, tmo.total_seconds()))
        if rehearsal:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        say("process group up")

    nb = K + W
    ins_per_step = 0 if CONFIG == 5 else D_PER_STEP // 10
    # Table size. Measured (profiles/r02_paths_and_load_factor.json): with 10M..15M resident rows the merge is fastest on a table of
    # ~44M slots (load factor 0.23-0.35); higher load factors lengthen the probe chains faster than the smaller table helps.
    n_profiled = min(K, 12) + min(K, 8)
    cap = int(os.environ.get("BMX_BENCH_CAP", max(22_000_000, R_PER_GPU + (nb + n_profiled + 3) * ins_per_step + 4 * D_PER_STEP)))
    eng = bmx.Engine(capacity_rows=cap, device=local_rank, load_pct=int(os.environ.get("BMX_BENCH_LOAD_PCT", 0)))
    # deferred compaction (include/bmx.h): on by default in the library; --no-defer is the A/B switch of the unsharded run. The sharded pipeline's choice is
    # bmx.sharded.EngineOps' (off unless BMX_SHARDED_DEFER=1: measured slower beside the exchange kernels); the bench only reports it.
    if not (world > 1 or args.force_sharded):
        eng.set_deferred(not args.no_defer)
    main_kernel = "k_probe_apply"
    verified = None
    refused_why = None

    if not sharded:
        rid = gen_resident(R_PER_GPU)
        eng.load_rows(*rid)
        host_batches = [gen_batch(b, R_PER_GPU) for b in range(nb)]
        batches = [to_dev(hb, dev) for hb in host_batches]
        applied = torch.zeros((nb, D_PER_STEP), dtype=torch.int32, device=dev)     # every batch keeps its own winner list: compared after the run
        n_applied = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()

        def step(b):
            i, f, t, v = batches[b]
            eng.merge_batch_dev(D_PER_STEP, i, f, t, v, bmx.INSERT_REFERENCE, applied=applied[b], n_applied=n_applied[b:b + 1])

        for b in range(W):
            step(b)
        eng.sync(); torch.cuda.synchronize()
        # The timed region: exactly K merge calls between two device-wide synchronisations. Inside the bracket only enqueue calls: the start event, the K
        # merges, and the stop event (timer_mark launches the last batch's still-pending compaction in front of it); torch.cuda.synchronize() waits for every
        # stream of the device, the engine's included. The engine's own status check (eng.sync: a device-to-host copy and a second wait) comes after the clock
        # is read — rounds 1-3 had it, and a blocking event wait, inside the bracket: ~5 us per step of host latency at K = 20 that is not merge time.
        t0 = time.perf_counter()
        eng.timer_start()
        for b in range(W, nb):
            step(b)
        eng.timer_mark()
        t_enq = time.perf_counter() - t0             # the host's share: K enqueue calls (a host-bound run shows here, not in the kernels)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ev_ms = eng.timer_elapsed()
        eng.sync()
        elapsed = wall
        winners = n_applied[W:nb].cpu().numpy()

        if not args.no_verify:
            verified = verify_against_oracle(eng, host_batches, applied, n_applied[:nb], rid)
        del rid, host_batches

        # second pass over fresh batches of the same shape: per-kernel HIP-event timing (live roofline figure)
        pbatches = [to_dev(gen_batch(nb + b, R_PER_GPU), dev) for b in range(min(K, 12))]
        torch.cuda.synchronize()
        eng.profile_enable(True)
        for (i, f, t, v) in pbatches:
            eng.merge_batch_dev(D_PER_STEP, i, f, t, v, bmx.INSERT_REFERENCE, applied=applied[0], n_applied=n_applied[nb:nb + 1])
        stage_ms, ncalls = eng.profile_read()
        extra = {}
        if CONFIG == 2:
            # same pass once more in the opt-in mode where the CALLER guarantees unique keys (no claim atomic, no resolve pass)
            ubatches = [to_dev(gen_batch(nb + 40 + b, R_PER_GPU), dev) for b in range(min(K, 8))]
            torch.cuda.synchronize()
            eng.profile_enable(True)
            eng.sync(); eng.timer_start()
            for (i, f, t, v) in ubatches:
                eng.merge_batch_dev(D_PER_STEP, i, f, t, v, bmx.INSERT_REFERENCE | bmx.MERGE_UNIQUE_KEYS, applied=applied[0], n_applied=n_applied[nb:nb + 1])
            ums = eng.timer_stop() / len(ubatches)
            ustage_ms, _ = eng.profile_read()
            extra["unique_keys_mode"] = {"note": "opt-in BMX_MERGE_UNIQUE_KEYS (caller guarantees no duplicate keys in the batch); not the headline",
                                         "ms_per_step_with_event_brackets": round(ums, 5), "kernel_ms": {k: round(v, 5) for k, v in ustage_ms.items()}}
        eng.profile_enable(False)
        wavg = float(winners.mean()) if len(winners) else 0.0
        # algorithmic bytes of one launch of the dominant kernel: delta read 28*D + resident row read 28*D
        # + (ts,val) store of the winners 16*W. (The 4*W index write belongs to the compaction launches.)
        alg_bytes = 56.0 * D_PER_STEP + 16.0 * wavg
        probe_s = stage_ms["probe_apply"] * 1e-3
        achieved = alg_bytes / probe_s / 1e9 if probe_s > 0 else 0.0
        traffic = requests = None
        tpath = os.path.join(ROOT, "profiles", "traffic_probe_apply.json")
        if os.path.exists(tpath) and CONFIG == 2 and main_kernel == "k_probe_apply":  # HBM bytes per launch from a committed rocprofv3 --pmc run of this same command
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("bytes_per_launch")
                requests = {"total": tj.get("requests_per_launch"), "reads": tj.get("read_requests"), "writes_incl_atomics": tj.get("write_requests"), "atomics": tj.get("atomic_requests")}
            except Exception:
                traffic = requests = None
        roofline = {"bound": "hbm", "kernel": main_kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "requests_per_launch": requests, "algorithmic_bytes_per_launch": alg_bytes,
                    "kernel_ms": {k: round(v, 5) for k, v in stage_ms.items()}, "launches_averaged": ncalls,
                    "whole_merge_achieved_GBs": round((56.0 * D_PER_STEP + 20.0 * wavg) / (elapsed / K) / 1e9, 1)}
        total_units = K * D_PER_STEP
        extra.update({"event_ms_per_step": round(ev_ms / K, 5), "host_enqueue_ms_per_step": round(t_enq / K * 1e3, 5), "winners_per_step": round(wavg, 1),
                      "table_placement": dict(eng.placement(), note="bmx_create allocates a large table several times and keeps the candidate on which the merge kernel's request mix (2^20 random slot reads + exchanges + stores) runs fastest: include/bmx.h bmx_get_placement"),
                      "deferred_compaction": dict(zip(("merges_deferred", "compactions_on_side_stream"), eng.deferred_counts()))})
        if CONFIG == 5:
            cfg = {"workload": "config 5: streaming sync replay on 1 MI355X, %d x 1M-delta batches (30%% of a batch on R/1000 = %d hot keys, 70%% uniform), steady state over batches %d..%d" %
                   (nb, R_PER_GPU // 1000, W, nb - 1), "resident_rows_per_gpu": R_PER_GPU, "deltas_per_step_per_gpu": D_PER_STEP, "insert_mode": "reference", "sharding": "none"}
        else:
            cfg = {"workload": "config 2: 10M-row resident graph on 1 MI355X, 1M-delta batch merge (90% hits / 10% inserts, unique keys in batch)",
                   "resident_rows_per_gpu": R_PER_GPU, "deltas_per_step_per_gpu": D_PER_STEP, "insert_mode": "reference", "sharding": "none"}
    else:
        from bmx.sharded import ShardedGraph, EngineOps
        if world > 1 and os.environ.get("BMX_SHARDED_EXCHANGE", "auto") == "auto":
            # every rank gets the same answer (all-reduce inside). A passed self-test leaves the choice on "auto": should the set-up of the REAL
            # slabs fail on some rank after all, every rank falls back to the RCCL exchange together instead of raising
            say("exchange self-test")
            st_kind, refused_why = exchange_selftest(dev, dist, rank, world)
            say("exchange self-test -> %s%s" % (st_kind, (" (direct refused: %s)" % refused_why) if refused_why else ""))
            if st_kind == "rccl":
                os.environ["BMX_SHARDED_EXCHANGE"] = "rccl"
        eng.set_wait_limit(WAIT_LIMIT_S)
        sg = ShardedGraph(EngineOps(eng, dev), dist, rank, world)
        if args.no_defer:
            eng.set_deferred(False)
        say("loading this rank's shard (%d of %d rows)" % (R_PER_GPU, R_PER_GPU * world))
        sg.load_owned_resident(R_PER_GPU, T0=T0, DT=DT)
        R_global = R_PER_GPU * world
        batches = [to_dev(gen_batch(b, R_global, seed=2 + 1000 * rank, part=(rank, world)), dev) for b in range(nb)]   # config 2: disjoint rows per originator
        say("%d batches generated; setting up the exchange" % nb)
        sg.setup_pipeline(D_PER_STEP, partition_on=os.environ.get("BMX_BENCH_PARTITION", "merge"), slack=1.25 if CONFIG == 5 else 1.03)
        torch.cuda.synchronize()
        say("exchange: %s; warm-up (%d steps)" % (sg.exchange, W))

        step_events = [] if os.environ.get("BMX_BENCH_STEPTIMES") else None   # debugging aid: when each step's merge finished (stderr)

        def run(lo, hi):
            # exactly (hi-lo) routes and (hi-lo) merges. route(b+1) is enqueued before merge(b): its partition runs on the merge
            # stream ahead of merge(b), its all-to-all on the communication stream underneath merge(b)
            tk = sg.route(D_PER_STEP, *batches[lo], exchange_now=True)   # nothing to hide the first exchange behind: start it before the next partition
            for b in range(lo, hi):
                nxt = sg.route(D_PER_STEP, *batches[b + 1]) if b + 1 < hi else None
                sg.merge(tk)
                if step_events is not None:
                    ev = torch.cuda.Event(enable_timing=True); ev.record(sg.ops.main); step_events.append(ev)
                tk = nxt

        if W:
            run(0, W)
        sg.ops.sync(); torch.cuda.synchronize(); dist.barrier()
        say("timed region (%d steps)" % K)
        t0 = time.perf_counter()
        if os.environ.get("BMX_BENCH_HOSTPROF"):      # where the host's enqueue time goes (stderr): wall time inside each call
            acc, each = {}, {}

            def timed(obj, name):
                fn = getattr(obj, name)

                def w(*a, **k):
                    t = time.perf_counter()
                    r = fn(*a, **k)
                    dt = time.perf_counter() - t
                    acc[name] = acc.get(name, 0.0) + dt
                    each.setdefault(name, []).append(dt)
                    return r
                setattr(obj, name, w)
            for nm in ("partition_slabs", "partition_slabs_on_comm", "merge_records", "signal", "wait_seq"):
                timed(sg.ops, nm)
            timed(dist, "all_to_all_single")
            if sg.exchange == "direct":       # the direct exchange's three calls per step
                timed(sg.ops.pe, "partition_scatter_raw"); timed(sg.ops.e, "merge_tail_wait"); timed(sg.ops.e, "merge_records_after")
            ttot = time.perf_counter()
            run(W, nb)
            acc["(whole loop)"] = time.perf_counter() - ttot
            print("host us/step: " + ", ".join("%s %.1f" % (k, v / K * 1e6) for k, v in sorted(acc.items())), file=sys.stderr)
            print("host us per call (median / max): " + ", ".join("%s %.1f / %.1f" % (k, float(np.median(v)) * 1e6, max(v) * 1e6) for k, v in sorted(each.items())), file=sys.stderr)
        else:
            run(W, nb)
        t_enq = time.perf_counter() - t0
        sg.ops.sync(); torch.cuda.synchronize(); dist.barrier()
        wall = time.perf_counter() - t0
        if step_events:
            te = step_events[-K:]
            print("step end-to-end gaps (us): " + " ".join("%.0f" % (te[i].elapsed_time(te[i + 1]) * 1e3) for i in range(len(te) - 1)) +
                  " | first timed step ends %.0f us after the last warm-up step" % (step_events[-K - 1].elapsed_time(te[0]) * 1e3 if len(step_events) > K else -1), file=sys.stderr)
        if sg.overflowed():
            raise SystemExit("exchange slab overflow: run invalid (raise ShardedGraph.setup_pipeline slack)")
        tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        say("timed region done: %.1f us per step here, %.1f max over ranks; verifying" % (wall / K * 1e6, elapsed / K * 1e6))
        total_units = K * D_PER_STEP * world
        if not args.no_verify and world * nb <= 256:
            # every rank replays, through the CPU oracle, the deltas of ALL originators that it owns (same generator, same order:
            # step by step, originators in rank order) and compares its shard's row count and digest
            from oracle.oracle import Oracle, rows_digest
            from bmx import synth
            tv = time.perf_counter()
            o = Oracle()
            o.load_rows(*sg.owned_resident_host(R_PER_GPU, T0=T0, DT=DT))
            for b in range(nb):
                for src in range(world):
                    cols = gen_batch(b, R_global, seed=2 + 1000 * src, part=(src, world))
                    mine = synth.owner_of_np(cols[0], world) == rank
                    o.merge_batch(*[c[mine] for c in cols])
            ok = eng.row_count() == len(o) and rows_digest(*eng.dump_rows()) == o.digest()
            per_rank = [None] * world
            dist.all_gather_object(per_rank, {"rank": rank, "ok": bool(ok), "rows": len(o), "table_digest": "%016x" % o.digest()})
            if not all(x["ok"] for x in per_rank):
                raise SystemExit("VERIFICATION FAILED: a shard's rows differ from the oracle replay (ranks: %s)" % ", ".join("%d %s" % (x["rank"], "ok" if x["ok"] else "MISMATCH") for x in per_rank))
            verified = {"against": "oracle/bmx_oracle.c: every rank replays the deltas it owns (all originators, step order) and compares its shard's row count and digest",
                        "batches": nb * world, "rows_this_rank": len(o), "ok": True, "per_rank": per_rank, "seconds": round(time.perf_counter() - tv, 2)}
            o.close()
        # second pass (every rank, same number of collectives): a few more steps with the per-kernel HIP-event brackets on, for this
        # rank's live k_probe_apply figure. The brackets are event records, i.e. stream bubbles: never part of the timed region.
        npro = min(K, 8)
        pb = [to_dev(gen_batch(nb + b, R_global, seed=2 + 1000 * rank, part=(rank, world)), dev) for b in range(npro)]
        torch.cuda.synchronize()
        eng.profile_enable(True)
        won = []
        for b in range(npro):
            p = sg.merge(sg.route(D_PER_STEP, *pb[b]))
            sg.ops.sync()
            won.append(int(p["n_applied"].cpu()[0]))
        stage_ms, ncalls = eng.profile_read()
        eng.profile_enable(False)
        wavg = float(np.mean(won)) if won else 0.0
        alg_bytes = 56.0 * D_PER_STEP + 16.0 * wavg         # same accounting as the N=1 line; padding records move no row bytes
        probe_s = stage_ms["probe_apply"] * 1e-3
        achieved = alg_bytes / probe_s / 1e9 if probe_s > 0 else 0.0
        traffic = None
        try:      # the same kernel on the same 1M-record batch shape: HBM bytes per launch of the committed rocprofv3 --pmc passes over the N=1 run
            traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic_probe_apply.json"))).get("bytes_per_launch")
        except Exception:
            traffic = None
        roofline = {"bound": "hbm", "kernel": main_kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                    "kernel_ms": {k: round(v, 5) for k, v in stage_ms.items()}, "launches_averaged": ncalls,
                    "traffic_source": "profiles/traffic_probe_apply.json: PMC passes over the N=1 run of this kernel (config 2, SoA columns), not re-collected per rank",
                    "note": "rank 0, launches without a concurrent exchange (second pass, per-kernel HIP events on the merge stream)"}
        how = ("direct: the owner partition of batch b+1 stores every slab straight into its owner's IPC-mapped receive memory (peer stores over xGMI) and sets arrival words; no collective, one stream"
               if sg.exchange == "direct" else "rccl: partition + ONE all-to-all of batch b+1 on a second stream under the merge of batch b")
        if sg.exchange != "direct" and refused_why is None:
            refused_why = getattr(sg, "direct_refused", None) or ("BMX_SHARDED_EXCHANGE=rccl" if os.environ.get("BMX_SHARDED_EXCHANGE") == "rccl" else None)
        extra = {"host_enqueue_ms_per_step": round(t_enq / K * 1e3, 4), "exchange": dict(sg.stats(), kind=sg.exchange, refused=(None if sg.exchange == "direct" else "direct"), why=refused_why, mode="fixed slabs of %d records per ordered pair; %s" % (sg.slab, how))}
        shape = ("config 5 shape: streaming replay with 30%% of every batch on %d global hot keys" % (R_global // 1000)) if CONFIG == 5 else "config 4 shape"
        cfg = {"workload": "%s: %dM-row graph id-hash sharded over %d MI355X, %dM mixed-shard deltas per step routed by RCCL all-to-all" %
               (shape, R_global // 1_000_000, world, world * D_PER_STEP // 1_000_000) if sg.exchange != "direct" else
               "%s: %dM-row graph id-hash sharded over %d MI355X, %dM mixed-shard deltas per step routed to their owners by direct peer stores (fallback: RCCL all-to-all)" %
               (shape, R_global // 1_000_000, world, world * D_PER_STEP // 1_000_000),
               "resident_rows_per_gpu": R_PER_GPU, "deltas_per_step_per_gpu": D_PER_STEP, "insert_mode": "reference", "sharding": "owner = hash(node id) mod N"}

    out = None
    if rank == 0:
        out = {"metric": "CRDT field-merges/s", "value": total_units / elapsed, "unit": "merges/s", "n_gpus": world, "steps": K, "warmup": W,
               "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64",
               "data": "synthetic", "config": cfg, "roofline": roofline, "verified": verified}
        out.update(extra)
        if not sharded and not args.no_scan:
            out["scan_config3"] = {}
            for rs in [int(x) for x in args.scan_rows.split(",") if x]:
                out["scan_config3"]["%dM" % (rs // 1_000_000)] = scan_bench(bmx, dev, R=rs)
                out["scan_config3"]["%dM_int64" % (rs // 1_000_000)] = scan_bench(bmx, dev, R=rs, wide=True)
        if not args.no_cpu_baseline:
            if not sharded:
                out["js_host"] = js_host_rate()
                out["cpu_baseline"] = cpu_baseline()
            else:   # the same one-core leg as the N=1 line, on rank 0, after the timed region (the other ranks wait at the barrier below)
                out["cpu_baseline"] = cpu_baseline(n_batches=12, extras=False)
        else:
            out["cpu_baseline"] = None
    if sharded:
        sg.close()
        sg.ops.close()
    eng.close()
    eng = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        emit(out, real_stdout)


if __name__ == "__main__":
    main()
