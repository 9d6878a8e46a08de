#!/bin/bash
# SQ counters of the exact-mode GEMM (separate passes, kernel-trace only).  usage: KPAT=gemm_q4_exact_h16 pmc_gx.sh tag qt M K N reps which
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU"
P3="SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_INSTS_SMEM"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P -d $R/gpurun_out/pmc_${tag}_$i -o out --output-format csv -- python $R/scripts/dev/gx_one.py "$@" > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for i in (1,2,3):
    fs = glob.glob("$R/gpurun_out/pmc_${tag}_%d/**/*counter_collection.csv" % i, recursive=True)
    if not fs: print("pass", i, "no csv"); continue
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "${KPAT:-gemm_q4_exact_mfma}" in r["Kernel_Name"]: a[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(a.items()): print(f"{k:32s} {sum(v)/len(v):16.0f}   (n={len(v)})")
PY
