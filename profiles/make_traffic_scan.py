"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --scan-rows 100000000` into profiles/traffic_scan.json:
HBM bytes per launch of k_scan_mask and k_scan_emit for each query of bench.py's scan_bench (two columns x four queries, 31 scans each, in
the order bench.py issues them), next to the algorithmic bytes. FETCH_SIZE is doubled as for the merge kernel (gfx950 tallies 128-B reads at 64 B).
usage: python profiles/make_traffic_scan.py <dir with pass_fetch/ pass_write/> <out.json>"""
import csv, glob, json, os, sys

QUERIES = ["equals_0.1pct", "range_1pct", "range_10pct", "range_50pct"]
PER_QUERY = 3 + 20 + 8          # warm-up, timed, per-kernel-event launches of scan_bench
R = 100_000_000
SEL = {"equals_0.1pct": 0.001, "range_1pct": 0.01, "range_10pct": 0.10, "range_50pct": 0.50}


def launches(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and ("k_scan_mask" in r["Kernel_Name"] or "k_scan_emit" in r["Kernel_Name"]):
                rows.append((int(r["Dispatch_Id"]), "mask" if "k_scan_mask" in r["Kernel_Name"] else "emit", float(r["Counter_Value"])))
    rows.sort()
    return rows


root, dst = sys.argv[1], sys.argv[2]
fe, wr = launches(os.path.join(root, "pass_fetch"), "FETCH_SIZE"), launches(os.path.join(root, "pass_write"), "WRITE_SIZE")
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) on `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                 "--no-verify --scan-rows 100000000`; profiles/make_traffic_scan.py",
       "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B); WRITE_SIZE exact", "rows": R, "queries": {}}
for kind, rows_f, rows_w in (("mask", [x for x in fe if x[1] == "mask"], [x for x in wr if x[1] == "mask"]),
                             ("emit", [x for x in fe if x[1] == "emit"], [x for x in wr if x[1] == "emit"])):
    assert len(rows_f) == len(rows_w) == 2 * len(QUERIES) * PER_QUERY, (kind, len(rows_f), len(rows_w))
    for ci, (col, w) in enumerate((("int32", 4), ("int64", 8))):
        for qi, q in enumerate(QUERIES):
            lo = (ci * len(QUERIES) + qi) * PER_QUERY
            f = [x[2] for x in rows_f[lo:lo + PER_QUERY]]
            ww = [x[2] for x in rows_w[lo:lo + PER_QUERY]]
            rd, wb = sum(f) / len(f) * 1024 * 2, sum(ww) / len(ww) * 1024
            e = out["queries"].setdefault("%s/%s" % (col, q), {"algorithmic_bytes": {"mask": w * R, "emit_ids_out": round(8 * R * SEL[q])}})
            e[kind] = {"read_bytes": round(rd), "write_bytes": round(wb), "bytes_per_launch": round(rd + wb), "launches_averaged": len(f)}
json.dump(out, open(dst, "w"), indent=1)
print(open(dst).read())
