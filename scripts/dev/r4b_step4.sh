#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_exact_gpu.py -m gpu -x -q -k "feed_forward or decode or model or graph" > gpurun_out/t_step4.txt 2>&1; tail -3 gpurun_out/t_step4.txt
for i in 1 2; do
echo "== pair2 (exchange)"; timeout 300 python scripts/exact_perf.py --eval 2>&1 | tail -1
echo "== FL_EXACT_PAIR1=1"; FL_EXACT_PAIR1=1 timeout 300 python scripts/exact_perf.py --eval 2>&1 | tail -1
done
