"""Print a window of a rocprofv3 kernel trace (CSV) as a per-queue timeline: python trace_timeline.py <kernel_trace.csv> [steps_from_end] [n_steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 2
idx = [i for i, r in enumerate(rows) if 'k_probe_apply' in r['Kernel_Name']]
s, e = idx[-back], idx[-back + cnt]
t0 = int(rows[s]['Start_Timestamp'])
for r in rows[s:e + 1]:
    nm = r['Kernel_Name'].split('(')[0].replace('bmx::', '').replace('void ', '')[:34]
    print("%-36s q%s start %8.1f end %8.1f dur %7.1f" % (nm, r['Queue_Id'], (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3,
                                                       (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
