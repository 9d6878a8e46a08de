#!/bin/bash
# round-4 evidence run on the GPU box: bench (default command), rocprofv3 --stats of the bench command, PMC traffic passes (separate --pmc
# passes, kernel-trace only), SQ counters of the reference-order GEMM.  usage: run_r4_final.sh [skip-tests]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_r4_final.json 2> gpurun_out/bench_r4_final.err; cut -c1-400 gpurun_out/bench_r4_final.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench_r4 -o out --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/prof_bench_r4.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc_pre_$c -o out --output-format csv -- python $GRAFT_REPO_ROOT/scripts/prefill_only.py 2 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc_dec_$c -o out --output-format csv -- python $GRAFT_REPO_ROOT/scripts/decode_only.py 8 0 > /dev/null 2>&1
  FL_FAST=1 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fpre_$c -o out --output-format csv -- python $GRAFT_REPO_ROOT/scripts/prefill_only.py 2 > /dev/null 2>&1
  FL_FAST=1 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fdec_$c -o out --output-format csv -- python $GRAFT_REPO_ROOT/scripts/decode_only.py 8 0 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
KPAT=gemm_q4_exact_h16 bash scripts/dev/pmc_gx.sh xh4 2 12288 4096 512 3 5 > gpurun_out/xh4_pmc.txt 2>&1
ls gpurun_out | grep "pmc_\|prof_bench"
