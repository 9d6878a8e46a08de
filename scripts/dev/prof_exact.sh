#!/bin/bash
# rocprofv3 kernel stats of the default-mode (reference-order) prefill and decode: usage prof_exact.sh tag
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=${1:-x}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_pre_$tag -o out --output-format csv -- python $R/scripts/prefill_only.py 4 > $R/gpurun_out/prof_pre_$tag.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dec_$tag -o out --output-format csv -- python $R/scripts/decode_only.py 32 0 > $R/gpurun_out/prof_dec_$tag.log 2>&1
cd $R
tail -2 gpurun_out/prof_pre_$tag.log; python scripts/dev/stats_summary.py gpurun_out/prof_pre_$tag | head -24
tail -3 gpurun_out/prof_dec_$tag.log; python scripts/dev/stats_summary.py gpurun_out/prof_dec_$tag | head -16
