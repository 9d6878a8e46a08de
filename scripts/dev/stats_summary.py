"""Summarise a rocprofv3 kernel-trace CSV: per kernel (template args kept, parameter lists dropped) calls / total / avg.
usage: stats_summary.py <dir-with-*kernel_trace.csv> [skip_first_n_dispatches_of_each_kernel]"""
import csv, glob, sys, collections, re
d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
a = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    a[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in a.values())
print(f"{'kernel':90s} {'calls':>7s} {'total ms':>10s} {'avg us':>9s} {'%':>6s}")
for k, v in sorted(a.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[:90]:90s} {len(v):7d} {sum(v)/1e3:10.3f} {sum(v)/len(v):9.2f} {100*sum(v)/tot:6.2f}")
