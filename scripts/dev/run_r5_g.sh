#!/bin/bash
# round 5, GPU run G: what the prologue's arithmetic costs a decode token (timing-only diagnostic build)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "default:FL_X=1" "prologue-compute-skipped:FASTLLAMA_HIP_LIB=gpurun_variants/libdiag.so" "default-again:FL_X=1"; do
  n=${v%%:*}; e=${v#*:}
  env $e python scripts/decode_only.py 64 1 0 128 2>&1 | grep decode | sed "s/^/[$n] /"
done
