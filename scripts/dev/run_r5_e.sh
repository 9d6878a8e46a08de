#!/bin/bash
# round 5, GPU run E: V.P pipelining A/B at deep contexts + kernel stats of a deep-context prefill
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "pipelined:FL_X=1" "round4-order:FASTLLAMA_HIP_LIB=gpurun_variants/libnopipe.so" "pipelined-again:FL_X=1"; do
  n=${v%%:*}; e=${v#*:}
  env $e python scripts/dev/prefill_deep.py 7B 512,1024,1536 2>&1 | grep n_past | sed "s/^/[$n] /"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5e_prof_deep -o out --output-format csv -- python $GRAFT_REPO_ROOT/scripts/dev/prefill_deep.py 7B 1536 > $GRAFT_REPO_ROOT/gpurun_out/r5e_prof_deep.log 2>&1
cd $GRAFT_REPO_ROOT; python scripts/dev/stats_summary.py gpurun_out/r5e_prof_deep | head -24
