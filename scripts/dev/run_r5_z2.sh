#!/bin/bash
# round 5, GPU run Z2: the batched rest loads of the Q8_0 prologue against the commit before, alternating on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do
  FASTLLAMA_HIP_LIB=$PWD/gpurun_variants/libprev.so python scripts/dev/decode_ab_lib.py 32 1 0 128 65B 2>&1 | tail -1 | sed "s/^/[before] /"
  python scripts/decode_only.py 32 1 0 128 65B 2>&1 | tail -1 | sed "s/^/[batched] /"
done
for i in 1 2; do
  FASTLLAMA_HIP_LIB=$PWD/gpurun_variants/libprev.so python scripts/dev/decode_ab_lib.py 48 1 0 128 13B 2>&1 | tail -1 | sed "s/^/[before] /"
  python scripts/decode_only.py 48 1 0 128 13B 2>&1 | tail -1 | sed "s/^/[batched] /"
done
