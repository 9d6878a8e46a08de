#!/bin/bash
# round-2 validation batch on the GPU box: test suite (minus the 7B parity runs), bench, PMC traffic passes, decode profile
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_parity_7b_gpu.py > gpurun_out/pytest_r2d.log 2>&1
tail -5 gpurun_out/pytest_r2d.log
python bench.py > gpurun_out/bench_r2b.json 2> gpurun_out/bench_r2b.err; tail -c 3000 gpurun_out/bench_r2b.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_pre_$C -o out --output-format csv -- python $R/scripts/prefill_only.py 2 > $R/gpurun_out/pmc_pre_$C.log 2>&1
  rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_dec_$C -o out --output-format csv -- python $R/scripts/decode_only.py 8 0 > $R/gpurun_out/pmc_dec_$C.log 2>&1
done
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_dec_r2a -o out --output-format csv -- python $R/scripts/decode_only.py 32 0 > $R/gpurun_out/prof_dec_r2a.log 2>&1
cd $R; python scripts/dev/stats_summary.py gpurun_out/prof_dec_r2a | head -14
