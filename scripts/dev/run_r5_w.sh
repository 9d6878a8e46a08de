#!/bin/bash
# round 5, GPU run W: the bench line once more on another box (the closing run's box measured 4-5 % below the earlier ones in every leg)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py > gpurun_out/r5w_bench.json 2> gpurun_out/r5w_bench.err; tail -2 gpurun_out/r5w_bench.err; cut -c1-260 gpurun_out/r5w_bench.json
python scripts/prefill_only.py 8 512 0 2>&1 | grep prefill
