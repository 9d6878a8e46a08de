#!/bin/bash
# rocprofv3 kernel stats of the FAST-mode decode (for the per-shape GEMV table): usage prof_fast_decode.sh tag
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=${1:-x}
FL_FAST=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fdec_$tag -o out --output-format csv -- python $R/scripts/decode_only.py 32 0 > $R/gpurun_out/prof_fdec_$tag.log 2>&1
cd $R; tail -3 gpurun_out/prof_fdec_$tag.log; python scripts/dev/stats_summary.py gpurun_out/prof_fdec_$tag | head -12
