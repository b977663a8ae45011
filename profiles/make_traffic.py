"""Turn three rocprofv3 --pmc passes over `python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-scan --no-verify` into profiles/traffic_probe_apply.json
(HBM bytes per k_probe_apply launch, corrected as MI355X_MICROARCH.md prescribes for gfx950).
usage: python profiles/make_traffic.py <dir with pass_fetch/ pass_write/ pass_req/ counter_collection CSVs> <out.json>"""
import csv, glob, json, os, sys

def means(d, kernel_sub):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if kernel_sub in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[k] = (sum(v) / len(v), len(v))
    return out

root, dst = sys.argv[1], sys.argv[2]
K = "k_probe_apply<false, 0, false"      # <AOS = false, BMX_INSERT_REFERENCE, UNIQUE = false[, NT]>
m = {}
for sub in ("pass_fetch", "pass_write", "pass_req"):
    m.update(means(os.path.join(root, sub), K))
fetch_kb, n = m["FETCH_SIZE"]; write_kb, _ = m["WRITE_SIZE"]
rd, _ = m["TCC_EA0_RDREQ_sum"]; rd128, _ = m["TCC_EA0_RDREQ_128B_sum"]; wr, _ = m["TCC_EA0_WRREQ_sum"]; at, _ = m["TCC_EA0_ATOMIC_sum"]
read_bytes = fetch_kb * 1024 * 2          # gfx950: FETCH_SIZE tallies 128-B read requests at 64 B
write_bytes = write_kb * 1024
json.dump({"kernel": "k_probe_apply<false,0,false,64>",
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_EA0_* (three separate passes) on `python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-scan --no-verify`; profiles/make_traffic.py",
           "launches_averaged": n, "FETCH_SIZE_KB_raw": round(fetch_kb, 1), "WRITE_SIZE_KB": round(write_kb, 1),
           "read_requests": round(rd), "read_requests_128B": round(rd128), "write_requests": round(wr), "atomic_requests": round(at),
           "correction": "gfx950 FETCH_SIZE tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM): x2; checked in round 1 on a 1 GiB copy (reports 524293 KB). WRITE_SIZE is exact (32-B partial write-backs + atomics counted as 32-B writes). requests_per_launch = read requests + write requests at the memory side (TCC_EA0_RDREQ + TCC_EA0_WRREQ; the write requests include the atomics, listed separately as atomic_requests).",
           "read_bytes": round(read_bytes), "write_bytes": round(write_bytes), "bytes_per_launch": round(read_bytes + write_bytes),
           "requests_per_launch": round(rd + wr)}, open(dst, "w"), indent=1)
print(open(dst).read())
