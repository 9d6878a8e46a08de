#!/bin/bash
# round 5, GPU run F: TEAMS form of the decode GEMV (tests, A/B, timeline), s_setprio around the lane-sum MFMAs of the prefill GEMM (A/B)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_exact_gpu.py tests/test_eval_ops_gpu.py -m gpu -x -q > gpurun_out/r5f_t1.txt 2>&1; tail -3 gpurun_out/r5f_t1.txt
FL_LLC_TEAMS_MIN=0 python -m pytest tests/test_exact_gpu.py tests/test_eval_ops_gpu.py -m gpu -x -q -k "gemv or pair or decode or model or forms" > gpurun_out/r5f_t2.txt 2>&1; tail -3 gpurun_out/r5f_t2.txt
for v in "teams:FL_X=1" "no-teams:FL_LLC_TEAMS=0" "teams-again:FL_X=1" "q41-teams:FL_QTYPE=3" "q41-no-teams:FL_QTYPE=3 FL_LLC_TEAMS=0"; do
  n=${v%%:*}; e=${v#*:}
  env $e python scripts/decode_only.py 64 1 0 128 2>&1 | grep decode | sed "s/^/[$n] /"
done
FASTLLAMA_HIP_LIB=gpurun_variants/libtl.so python scripts/dev/decode_timeline.py 7B 128 > gpurun_out/r5f_decode_timeline.md 2> gpurun_out/r5f_decode_timeline.err; tail -2 gpurun_out/r5f_decode_timeline.err; grep -v "^$" gpurun_out/r5f_decode_timeline.md | head -10
for v in "default:FL_X=1" "setprio:FASTLLAMA_HIP_LIB=gpurun_variants/libsetprio.so" "default-again:FL_X=1" "setprio-again:FASTLLAMA_HIP_LIB=gpurun_variants/libsetprio.so"; do
  n=${v%%:*}; e=${v#*:}
  env $e python scripts/prefill_only.py 8 2>&1 | grep prefill | sed "s/^/[$n] /"
done
