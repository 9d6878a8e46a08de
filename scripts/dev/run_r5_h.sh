#!/bin/bash
# round 5, GPU run H: TEAMS with the chain waves of a phase on different SIMDs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FL_LLC_TEAMS_MIN=0 python -m pytest tests/test_exact_gpu.py tests/test_eval_ops_gpu.py -m gpu -x -q -k "gemv or pair or decode or model or forms" > gpurun_out/r5h_t2.txt 2>&1; tail -3 gpurun_out/r5h_t2.txt
for v in "default:FL_X=1" "teams:FL_LLC_TEAMS=1" "default-again:FL_X=1" "teams-again:FL_LLC_TEAMS=1" "q41:FL_QTYPE=3" "q41-teams:FL_QTYPE=3 FL_LLC_TEAMS=1"; do
  n=${v%%:*}; e=${v#*:}
  env $e python scripts/decode_only.py 64 1 0 128 2>&1 | grep decode | sed "s/^/[$n] /"
done
FL_LLC_TEAMS=1 FASTLLAMA_HIP_LIB=gpurun_variants/libtl.so python scripts/dev/decode_timeline.py 7B 128 > gpurun_out/r5h_decode_timeline.md 2> gpurun_out/r5h_decode_timeline.err; tail -2 gpurun_out/r5h_decode_timeline.err; grep -v "^$" gpurun_out/r5h_decode_timeline.md | head -10
