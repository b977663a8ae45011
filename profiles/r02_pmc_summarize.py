"""Summarise the rocprofv3 --pmc passes of profiles/r02_pmc_passes.sh: per-launch means of every counter for the merge kernels.
usage: python profiles/r02_pmc_summarize.py <dir with pass*/ ... counter_collection.csv> <out.json>"""
import csv, glob, json, os, sys

root, dst = sys.argv[1], sys.argv[2]
KERNELS = {"k_probe_apply<false, 0, false>": "k_probe_apply", "k_resolve_lists<false, 0>": "k_resolve_lists", "k_compact_winners": "k_compact_winners"}
out = {v: {} for v in KERNELS.values()}
for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    acc = {}
    for r in csv.DictReader(open(f)):
        for sub, name in KERNELS.items():
            if sub in r["Kernel_Name"]:
                acc.setdefault((name, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (name, c), v in acc.items():
        out[name][c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
k = out["k_probe_apply"]
g = lambda c: k.get(c, {}).get("mean_per_launch")
d = {}
if g("TCC_EA0_RDREQ_sum") and g("TCC_EA0_RDREQ_LEVEL_sum"):
    d["avg_EA_read_latency_TCC_cycles"] = g("TCC_EA0_RDREQ_LEVEL_sum") / g("TCC_EA0_RDREQ_sum")
if g("TCC_EA0_WRREQ_sum") and g("TCC_EA0_WRREQ_LEVEL_sum"):
    d["avg_EA_write_latency_TCC_cycles"] = g("TCC_EA0_WRREQ_LEVEL_sum") / g("TCC_EA0_WRREQ_sum")
if g("TCC_EA0_ATOMIC_sum") and g("TCC_EA0_ATOMIC_LEVEL_sum"):
    d["avg_EA_atomic_latency_TCC_cycles"] = g("TCC_EA0_ATOMIC_LEVEL_sum") / g("TCC_EA0_ATOMIC_sum")
if g("TCC_CYCLE_sum"):
    cyc = g("TCC_CYCLE_sum")
    for c in ("TCC_BUSY_sum", "TCC_EA0_WRREQ_STALL_sum", "TCC_TOO_MANY_EA_WRREQS_STALL_sum", "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum",
              "TCC_TAG_STALL_sum", "TCC_SRC_FIFO_FULL_sum", "TCC_LATENCY_FIFO_FULL_sum", "TCC_IB_STALL_sum"):
        if g(c) is not None:
            d[c.replace("_sum", "") + "_frac_of_TCC_cycles"] = g(c) / cyc
if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum"):
    d["L2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY") is not None:
    d["SQ_WAIT_ANY_frac_of_wave_cycles"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
    if g("SQ_WAIT_INST_ANY") is not None: d["SQ_WAIT_INST_ANY_frac_of_wave_cycles"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
    if g("SQ_ACTIVE_INST_ANY") is not None: d["SQ_ACTIVE_INST_ANY_frac_of_wave_cycles"] = g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")
if g("TCP_TCC_READ_REQ_sum") and g("TCP_TCC_READ_REQ_LATENCY_sum"):
    d["avg_L1_to_L2_read_latency_cycles"] = g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum")
if g("TCP_TCC_WRITE_REQ_sum") and g("TCP_TCC_WRITE_REQ_LATENCY_sum"):
    d["avg_L1_to_L2_write_latency_cycles"] = g("TCP_TCC_WRITE_REQ_LATENCY_sum") / g("TCP_TCC_WRITE_REQ_sum")
out["derived_k_probe_apply"] = d
out["source"] = "profiles/r02_pmc_passes.sh: rocprofv3 --kernel-trace --pmc <group> over `python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline`, one group per pass"
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(d, indent=1))
