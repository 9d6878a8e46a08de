"""gaps between consecutive kernels of a rocprofv3 --kernel-trace csv (same queue, sorted by start): python scripts/dev/gap_summary.py <dir>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in csv.DictReader(open(f))]
rows.sort()
gaps = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = (s1 - e0) / 1e3
    if -5 < g < 20:
        gaps[(n0[:42], n1[:42])].append(g)
tot = 0
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:14]:
    v.sort()
    print(f"{len(v):6d} x gap med {v[len(v)//2]:6.2f} us  p10 {v[len(v)//10]:6.2f}  p90 {v[len(v)*9//10]:6.2f}   {k[0]} -> {k[1]}")
