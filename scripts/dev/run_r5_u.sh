#!/bin/bash
# round 5, GPU run U: multi-pass rows on by shape: per-kernel times of a 65B decode token, tests of the default path, TP rehearsal at 65B width
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r5u_prof -o out --output-format csv -- python $R/scripts/decode_only.py 16 0 0 128 65B > $R/gpurun_out/r5u_prof.log 2>&1
cd $R
python scripts/dev/stats_summary.py gpurun_out/r5u_prof > gpurun_out/r5u_decode65_kernel_stats.txt; head -14 gpurun_out/r5u_decode65_kernel_stats.txt; rm -rf gpurun_out/r5u_prof
tail -1 gpurun_out/r5u_prof.log
timeout 1500 python -m pytest tests/test_exact_gpu.py tests/test_wide_models_gpu.py tests/test_full_size_gpu.py -m gpu -x -q > gpurun_out/r5u_t1.txt 2>&1; tail -2 gpurun_out/r5u_t1.txt
FL_LAYERS=4 timeout 800 python scripts/dev/tp_decode_rehearsal.py 65B 8 /tmp/tpr65 64 2>&1 | grep "rank [01]" | sort | sed "s/^/[65B-width x 4 layers, G=8] /"
timeout 800 python scripts/dev/tp_decode_rehearsal.py 13B 2 /tmp/tpr13 64 2>&1 | grep "rank" | sort | sed "s/^/[13B, G=2] /"
