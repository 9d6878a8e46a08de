#!/bin/bash
# round 5, GPU run C: PERSIST instantiation + LX-direct prologue of the decode GEMV (tests, A/B), timeline, full-size parity, coexec6 SQ counters
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_exact_gpu.py tests/test_eval_ops_gpu.py -m gpu -x -q > gpurun_out/r5c_t1.txt 2>&1; tail -3 gpurun_out/r5c_t1.txt
FL_LLC_SLOTS=6 python -m pytest tests/test_exact_gpu.py tests/test_eval_ops_gpu.py -m gpu -x -q -k "gemv or pair or decode or model or forms" > gpurun_out/r5c_t2.txt 2>&1; tail -3 gpurun_out/r5c_t2.txt
for v in "default:FL_X=1" "nopersist:FL_LLC_NOPERSIST=1" "default-again:FL_X=1"; do
  n=${v%%:*}; e=${v#*:}
  env $e python scripts/decode_only.py 64 1 0 128 2>&1 | grep decode | sed "s/^/[$n] /"
done
FASTLLAMA_HIP_LIB=gpurun_variants/libtl.so python scripts/dev/decode_timeline.py 7B 128 > gpurun_out/r5c_decode_timeline.md 2> gpurun_out/r5c_decode_timeline.err; tail -2 gpurun_out/r5c_decode_timeline.err; grep -v "^$" gpurun_out/r5c_decode_timeline.md | head -12
python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -s > gpurun_out/r5c_fullsize.txt 2>&1; tail -8 gpurun_out/r5c_fullsize.txt
cd /tmp && export TMPDIR=/tmp
for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  t=$(echo $P | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $P -d $GRAFT_REPO_ROOT/gpurun_out/r5c_pmc_coexec6_$t -o out --output-format csv -- $GRAFT_REPO_ROOT/scripts/ubench/coexec6 4 > $GRAFT_REPO_ROOT/gpurun_out/r5c_pmc_coexec6_$t.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r5c_pmc_coexec6_*/")):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    a = collections.defaultdict(dict)
    for r in csv.DictReader(open(fs[0])):
        key = (r["Kernel_Name"][:40], r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")), r["Dispatch_Id"])
        a[key][r["Counter_Name"]] = float(r["Counter_Value"])
    for k, v in a.items(): print(k, {n: int(x) for n, x in v.items()})
PY
