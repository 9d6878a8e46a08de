#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_exact_gpu.py -m gpu -x -q > gpurun_out/t_step2.txt 2>&1; tail -4 gpurun_out/t_step2.txt
timeout 600 python -m pytest tests/test_wide_models_gpu.py -m gpu -x -q -k "row_split and 13B" > gpurun_out/t_step2b.txt 2>&1; tail -2 gpurun_out/t_step2b.txt
echo "== eval timings (pair2)"; timeout 300 python scripts/exact_perf.py --eval 2>&1 | tail -2
echo "== eval timings (FL_EXACT_PAIR1=1)"; FL_EXACT_PAIR1=1 timeout 300 python scripts/exact_perf.py --eval 2>&1 | tail -1
bash scripts/dev/prof_exact.sh r4c > gpurun_out/prof_r4c.txt 2>&1; grep "fl::" gpurun_out/prof_r4c.txt | grep -v "repack\|quantize_row\|to_h16\|to_qwd" | head -24
