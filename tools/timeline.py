"""Print the kernel timeline of one training step from a rocprofv3 --kernel-trace csv."""
import csv, glob, sys
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob('gpurun_out/prof_*/*/*kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step = from one fraction_load_kernel (the first kernel of a step) to the kernel in front of the next one; with the re-layout
# prefetched on the side stream (cn_fraction_prefetch_resident) name another once-per-step kernel as argv[2], e.g.
# softmax_mcc_bwd_kernel: the listing then runs from one backward pass to the next, the step boundary in its middle
delim = sys.argv[2] if len(sys.argv) > 2 else 'fraction_load_kernel'
sg = [i for i, r in enumerate(rows) if delim in r['Kernel_Name']]
m = int(len(sg) * (float(sys.argv[3]) if len(sys.argv) > 3 else 0.25))      # a step of bench.py's first, timed leg (host-buffer and check legs follow)
a, b = sg[m] - 1, sg[m + 1] - 1
t0 = int(rows[a]['End_Timestamp'])
prev_end = t0
agg = {}
for r in rows[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = r['Kernel_Name'].replace('void cn::', '').replace('cn::', '')[:40]
    print("%-42s q=%s start=%8.1f dur=%7.1f gap=%7.1f grid=%s" % (nm, r.get('Queue_Id', '?'), (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r['Grid_Size_X']))
    prev_end = max(prev_end, e)
    k = nm.split('(')[0].split('<')[0]
    agg[k] = agg.get(k, 0) + (e - s) / 1e3
print("step total us", (int(rows[b]['End_Timestamp']) - t0) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print("  %-30s %8.1f us" % (k, v))
