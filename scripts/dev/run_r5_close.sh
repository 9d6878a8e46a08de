#!/bin/bash
# closing run of round 5 on one box: the whole -m gpu suite, smoke(), the bench line, kernel stats of the bench command and of the reference-order
# prefill / decode, PMC traffic passes (separate --pmc passes, kernel-trace only)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r5_t_close.txt 2>&1; tail -3 gpurun_out/r5_t_close.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r5_t_close.txt
python bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err; tail -2 gpurun_out/r5_bench.err; cut -c1-300 gpurun_out/r5_bench.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r5_prof_bench -o out --output-format csv -- python $R/bench.py --no-cpu-baseline --no-other-configs > $R/gpurun_out/r5_prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r5_prof_pre -o out --output-format csv -- python $R/scripts/prefill_only.py 4 > $R/gpurun_out/r5_prof_pre.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r5_prof_dec -o out --output-format csv -- python $R/scripts/decode_only.py 32 0 > $R/gpurun_out/r5_prof_dec.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_pre_$c -o out --output-format csv -- python $R/scripts/prefill_only.py 2 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_dec_$c -o out --output-format csv -- python $R/scripts/decode_only.py 8 0 > /dev/null 2>&1
done
cd $R
python scripts/dev/stats_summary.py gpurun_out/r5_prof_bench > gpurun_out/r5_bench_kernel_stats.txt; head -12 gpurun_out/r5_bench_kernel_stats.txt
python scripts/dev/stats_summary.py gpurun_out/r5_prof_pre > gpurun_out/r5_prefill_kernel_stats.txt
python scripts/dev/stats_summary.py gpurun_out/r5_prof_dec > gpurun_out/r5_decode_kernel_stats.txt; head -10 gpurun_out/r5_decode_kernel_stats.txt
# the raw traces are too big to travel back (gpurun merges <= 64 MiB): summarise here, keep the summaries
ROUND=r05 PMC_OUT=gpurun_out python scripts/pmc_summary.py > /dev/null 2>&1; ls -la gpurun_out/r05_pmc_traffic.* 2>&1 | tail -2
rm -rf gpurun_out/r5_prof_bench gpurun_out/r5_prof_pre gpurun_out/r5_prof_dec gpurun_out/pmc_pre_* gpurun_out/pmc_dec_*
du -sh gpurun_out | tail -1
