#!/bin/bash
# round-3 final evidence run on the GPU box: whole -m gpu suite, smoke, bench, rocprofv3 --stats of the bench command, PMC passes
cd $GRAFT_REPO_ROOT
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/pytest_r3_final.log 2>&1
tail -5 gpurun_out/pytest_r3_final.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > gpurun_out/smoke_r3.log 2>&1; tail -2 gpurun_out/smoke_r3.log
python bench.py > gpurun_out/bench_r3_final.json 2> gpurun_out/bench_r3_final.err; cut -c1-600 gpurun_out/bench_r3_final.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench_r3 -o out --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fast --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/prof_bench_r3.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/prof_bench_r3 | head
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc_pre_$c -o out --output-format csv -- python $GRAFT_REPO_ROOT/scripts/prefill_only.py 2 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc_dec_$c -o out --output-format csv -- python $GRAFT_REPO_ROOT/scripts/decode_only.py 8 0 > /dev/null 2>&1
done
ls $GRAFT_REPO_ROOT/gpurun_out/ | grep pmc_
