#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_exact_gpu.py tests/test_model_gpu.py -m gpu -x -q > gpurun_out/t_step3.txt 2>&1; tail -3 gpurun_out/t_step3.txt
for i in 1 2; do
echo "== prefetch on"; timeout 300 python scripts/exact_perf.py --eval 2>&1 | tail -2
echo "== FL_NO_PREFETCH=1"; FL_NO_PREFETCH=1 timeout 300 python scripts/exact_perf.py --eval 2>&1 | tail -2
done
