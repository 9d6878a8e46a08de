#!/usr/bin/env python3
"""Compile one .hip file for gfx950 and print, per kernel: VGPRs, spills, occupancy, and an instruction histogram of the
hottest loop (the innermost loop with the most MFMAs).  Development aid.   usage: isa_summary.py file.hip [name-filter]"""
import collections, re, subprocess, sys, os, tempfile
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tmp = tempfile.mkdtemp()
r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Iinclude", "--cuda-device-only", "-S", src, "-o", tmp + "/dev-gfx950.s",
                    "-Rpass-analysis=kernel-resource-usage"] + [a for a in sys.argv[3:] if a != "--keep"], capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-4000:]); sys.exit(1)
res = {}
cur = None
for l in r.stderr.split("\n"):
    m = re.search(r"Function Name: (\S+)", l)
    if m: cur = m.group(1); res[cur] = {}
    for k in ("VGPRs:", "AGPRs:", "VGPRs Spill:", "SGPRs:", "Occupancy [waves/SIMD]:", "LDS Size [bytes/block]:"):
        m = re.search(re.escape(k) + r" (\d+)", l)
        if m and cur and k not in res[cur]: res[cur][k] = int(m.group(1))
asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
s = open(tmp + "/" + asm).read()
for name, rr in res.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"\(.*", "", dem)
    if flt and flt not in short: continue
    i0 = s.find("\n" + name + ":")
    if i0 < 0: continue
    body = s[i0:s.find(".Lfunc_end", i0)].split("\n")
    labels = {l.strip()[:-1].split(":")[0]: i for i, l in enumerate(body) if re.match(r"\.LBB\d+_\d+:", l.strip())}
    loops = []
    for i, l in enumerate(body):
        mm = re.match(r"\s*s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i: loops.append((labels[mm.group(1)], i))
    best, bc = None, None
    for a, b in loops:
        c = collections.Counter(x.split()[0] for x in body[a:b + 1] if x.strip() and not x.strip().startswith((".", ";")))
        nm = sum(v for k, v in c.items() if k.startswith("v_mfma"))
        if best is None or nm > best[0]: best, bc = (nm, a, b), c
    print(f"{short}\n   VGPR {rr.get('VGPRs:')} AGPR {rr.get('AGPRs:')} spill {rr.get('VGPRs Spill:')} occ {rr.get('Occupancy [waves/SIMD]:')}")
    if bc:
        g = lambda p: sum(v for k, v in bc.items() if k.startswith(p))
        valu = sum(v for k, v in bc.items() if k.startswith("v_") and not k.startswith("v_mfma"))
        print(f"   hot loop lines {best[1]}..{best[2]}: mfma {g('v_mfma')} valu {valu} salu {g('s_')} ds {g('ds_')} vmem {g('buffer_') + g('global_')} "
              f"scratch {g('scratch_')} waitcnt {bc['s_waitcnt']} nop {bc['s_nop']}")
        print("   valu:", ", ".join(f"{k}:{v}" for k, v in sorted(bc.items(), key=lambda x: -x[1]) if k.startswith("v_") and not k.startswith("v_mfma")))
if "--keep" in sys.argv: print(tmp + "/" + asm)
