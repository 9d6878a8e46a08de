#!/bin/bash
# round-4 closing evidence on the GPU box: bench (default command), kernel stats of the reference-order prefill / decode and of the bench
# command.  (PMC passes: scripts/dev/run_r4_final.sh.)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_r4_final.json 2> gpurun_out/bench_r4_final.err; cut -c1-300 gpurun_out/bench_r4_final.json
bash scripts/dev/prof_exact.sh r4f > gpurun_out/prof_r4f.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench_r4 -o out --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/prof_bench_r4.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_bench_r4.log | cut -c1-200
