#!/bin/bash
# The closing run of a round on ONE box (usage: gpurun -- 'bash scripts/dev/close_round.sh r06'): the whole -m gpu suite, smoke(), the bench line,
# rocprofv3 kernel stats of the bench command and of the reference-order prefill / decode, the PMC traffic passes (separate --pmc passes,
# kernel-trace only), the in-graph decode timelines (needs gpurun_variants/libtl.so: ALL_FLAGS=-DLLC_TIMING TAG=tl OUT=gpurun_variants/libtl.so
# bash scripts/dev/fastbuild.sh).  Summaries land in gpurun_out/<round>_*; copy the ones to keep into profiles/.
R=$GRAFT_REPO_ROOT; T=${1:-rXX}
cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/${T}_gpu_tests.txt 2>&1; tail -3 gpurun_out/${T}_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/${T}_gpu_tests.txt
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; tail -2 gpurun_out/${T}_bench.err; cut -c1-300 gpurun_out/${T}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o out --output-format csv -- python $R/bench.py --no-cpu-baseline --no-other-configs > $R/gpurun_out/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_pre -o out --output-format csv -- python $R/scripts/prefill_only.py 4 > $R/gpurun_out/prof_pre.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dec -o out --output-format csv -- python $R/scripts/decode_only.py 32 0 > $R/gpurun_out/prof_dec.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_pre_$c -o out --output-format csv -- python $R/scripts/prefill_only.py 2 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_dec_$c -o out --output-format csv -- python $R/scripts/decode_only.py 8 0 > /dev/null 2>&1
done
cd $R
python scripts/dev/stats_summary.py gpurun_out/prof_bench > gpurun_out/${T}_bench_kernel_stats.txt; head -12 gpurun_out/${T}_bench_kernel_stats.txt
python scripts/dev/stats_summary.py gpurun_out/prof_pre > gpurun_out/${T}_prefill_exact_kernel_stats.txt
python scripts/dev/stats_summary.py gpurun_out/prof_dec > gpurun_out/${T}_decode_exact_kernel_stats.txt; head -10 gpurun_out/${T}_decode_exact_kernel_stats.txt
# (the raw traces are too big to travel back -- gpurun merges <= 64 MiB: summarised here, the summaries travel)
ROUND=$T PMC_OUT=gpurun_out python scripts/pmc_summary.py > /dev/null 2>&1; ls -la gpurun_out/${T}_pmc_traffic.* 2>&1 | tail -2
rm -rf gpurun_out/prof_bench gpurun_out/prof_pre gpurun_out/prof_dec gpurun_out/pmc_pre_* gpurun_out/pmc_dec_*
if [ -f gpurun_variants/libtl.so ]; then
  FASTLLAMA_HIP_LIB=$R/gpurun_variants/libtl.so python scripts/dev/decode_timeline.py 7B 128 > gpurun_out/${T}_decode_timeline_llc.md 2> gpurun_out/${T}_tl.err; tail -3 gpurun_out/${T}_tl.err
  FASTLLAMA_HIP_LIB=$R/gpurun_variants/libtl.so python scripts/dev/stream_timeline.py 7B 128 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${T}_decode_timeline_stream.md
fi
du -sh gpurun_out | tail -1
