#!/bin/bash
# round 4, second half: tests of the reworked decode attention leftovers / two-workgroup w1|w3 pairs / fused P.V quantization, then A/B timings
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_exact_gpu.py tests/test_eval_ops_gpu.py -m gpu -x -q > gpurun_out/t_step1.txt 2>&1; tail -4 gpurun_out/t_step1.txt
timeout 600 python -m pytest tests/test_wide_models_gpu.py -m gpu -x -q -k "row_split and 13B" > gpurun_out/t_step1b.txt 2>&1; tail -2 gpurun_out/t_step1b.txt
echo "== eval timings (pair2)"; timeout 300 python scripts/exact_perf.py --eval 2>&1 | tail -4
echo "== eval timings (FL_EXACT_PAIR1=1)"; FL_EXACT_PAIR1=1 timeout 300 python scripts/exact_perf.py --eval 2>&1 | tail -2
bash scripts/dev/prof_exact.sh r4b > gpurun_out/prof_r4b.txt 2>&1; cat gpurun_out/prof_r4b.txt
