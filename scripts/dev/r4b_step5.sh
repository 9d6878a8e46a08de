#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_exact_gpu.py -m gpu -x -q > gpurun_out/t_step5.txt 2>&1; tail -3 gpurun_out/t_step5.txt
for i in 1 2; do
echo "== persistent"; timeout 300 python scripts/exact_perf.py --eval 2>&1 | tail -1
echo "== FL_XH_NOPERSIST=1"; FL_XH_NOPERSIST=1 timeout 300 python scripts/exact_perf.py --eval 2>&1 | tail -1
done
echo "== Q4_1 persistent"; timeout 300 python scripts/exact_perf.py --eval --qtype 3 2>&1 | tail -1
echo "== Q4_1 FL_XH_NOPERSIST=1"; FL_XH_NOPERSIST=1 timeout 300 python scripts/exact_perf.py --eval --qtype 3 2>&1 | tail -1
