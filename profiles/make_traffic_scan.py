"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --scan-rows 100000000` into profiles/traffic_scan.json:
HBM bytes per launch of the scan kernels for each query of bench.py's scan_bench (two columns x four queries, in the order bench.py issues them), next to
the algorithmic bytes: k_scan_mask (the one read of the value column), k_scan_emit with id output (gathers the id column) and with position output
(reads only the mask). FETCH_SIZE is doubled as for the merge kernel (gfx950 tallies 128-B reads at 64 B).
usage: python profiles/make_traffic_scan.py <dir with pass_fetch/ pass_write/> <out.json>"""
import csv, glob, json, os, sys

QUERIES = ["equals_0.1pct", "range_1pct", "range_10pct", "range_50pct"]
# launches per query in scan_bench: id output 3 warm-up + 20 timed, position output 3 + 20, count-only 20 (mask without the mask write, no emit), 8 between events (id output)
PER = {"mask": 23 + 23 + 8, "mask_count": 20, "emit": 23 + 8, "emit_pos": 23}
R = 100_000_000
SEL = {"equals_0.1pct": 0.001, "range_1pct": 0.01, "range_10pct": 0.10, "range_50pct": 0.50}


def kind_of(name):
    if "k_scan_mask" in name:
        return "mask" if ", true>" in name else "mask_count"
    if "k_scan_emit" in name:
        return "emit_pos" if "EmitPos" in name else "emit"
    return None


def launches(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = kind_of(r["Kernel_Name"])
            if r["Counter_Name"] == counter and k:
                rows.append((int(r["Dispatch_Id"]), k, float(r["Counter_Value"])))
    rows.sort()
    return rows


root, dst = sys.argv[1], sys.argv[2]
fe, wr = launches(os.path.join(root, "pass_fetch"), "FETCH_SIZE"), launches(os.path.join(root, "pass_write"), "WRITE_SIZE")
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) on `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                 "--no-verify --scan-rows 100000000`; profiles/make_traffic_scan.py",
       "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B); WRITE_SIZE exact", "rows": R, "queries": {}}
for kind, per in PER.items():
    rows_f, rows_w = [x for x in fe if x[1] == kind], [x for x in wr if x[1] == kind]
    # per column: the four queries' launches, then (id output only) the column scan that checks the view after its patches and the one of "first_scan_after_a_1M_delta_merge", which are left out
    assert len(rows_f) == len(rows_w) and len(rows_f) % 2 == 0 and len(rows_f) // 2 - len(QUERIES) * per in (0, 1, 2), (kind, len(rows_f), len(rows_w))
    per_col = len(rows_f) // 2
    for ci, (col, w) in enumerate((("int32", 4), ("int64", 8))):
        for qi, q in enumerate(QUERIES):
            lo = ci * per_col + qi * per
            f = [x[2] for x in rows_f[lo:lo + per]]
            ww = [x[2] for x in rows_w[lo:lo + per]]
            rd, wb = sum(f) / len(f) * 1024 * 2, sum(ww) / len(ww) * 1024
            e = out["queries"].setdefault("%s/%s" % (col, q), {"algorithmic_bytes": {"mask": w * R, "emit_ids_out": round(8 * R * SEL[q]), "emit_positions_out": round(4 * R * SEL[q])}})
            e[kind] = {"read_bytes": round(rd), "write_bytes": round(wb), "bytes_per_launch": round(rd + wb), "launches_averaged": len(f)}
json.dump(out, open(dst, "w"), indent=1)
print(open(dst).read())
