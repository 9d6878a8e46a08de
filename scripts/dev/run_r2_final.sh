#!/bin/bash
# round-2 final evidence run on the GPU box: whole -m gpu suite, smoke, bench, rocprofv3 --stats of the bench command
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_r2_final.log 2>&1
tail -4 gpurun_out/pytest_r2_final.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > gpurun_out/smoke_r2.log 2>&1; tail -2 gpurun_out/smoke_r2.log
python bench.py > gpurun_out/bench_r2_final.json 2> gpurun_out/bench_r2_final.err; cut -c1-400 gpurun_out/bench_r2_final.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench_r2 -o out --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench_r2.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/prof_bench_r2 | head
