"""Kernel timeline of the last full step of tools/split_probe.py (rocprofv3 --kernel-trace csv): all queues, sorted by start."""
import csv, glob, sys
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fl = [i for i, r in enumerate(rows) if 'fraction_load' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
a, b = fl[-2 * k], fl[-k]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = r['Kernel_Name'].replace('void cn::', '').replace('cn::', '')[:34]
    print("%-36s q=%-3s start=%8.1f dur=%7.1f" % (nm, r.get('Queue_Id', '?'), (s - t0) / 1e3, (e - s) / 1e3))
