#!/bin/bash
# round 5, GPU run L: the 8 x 8 form of the decode GEMV for K <= 8192 with many row groups (LLaMA-65B widths)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_exact_gpu.py -m gpu -x -q -k "gemv_with_norm or feed_forward" > gpurun_out/r5l_t1.txt 2>&1; tail -2 gpurun_out/r5l_t1.txt
for v in "llc-8x8:FL_X=1" "round3-kernel:FL_EXACT_R3=1"; do
  n=${v%%:*}; e=${v#*:}
  env FL_NCTX=1024 $e python scripts/decode_only.py 16 1 0 128 65B 2>&1 | grep decode | sed "s/^/[$n] /"
done
env FL_NCTX=1024 python scripts/decode_only.py 24 1 0 128 13B 2>&1 | grep decode | sed "s/^/[13B] /"
