"""median duration per kernel name of a rocprofv3 --kernel-trace csv: python scripts/dev/ktrace_summary.py <dir> [min_count]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(r["Kernel_Name"][:110], []).append(d)
mn = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= mn:
        v.sort()
        print(f"{len(v):6d} x med {v[len(v)//2]:8.2f} us  sum {sum(v)/1e3:9.3f} ms  {k}")
