#!/bin/bash
# round 5, GPU run X: does the exchange-tail parameter cost the single-GPU decode anything?  The library of the commit before it against HEAD, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3; do
  FASTLLAMA_HIP_LIB=$PWD/gpurun_variants/libold.so python scripts/dev/decode_ab_lib.py 96 1 0 128 2>&1 | tail -1 | sed "s/^/[before the tail] /"
  python scripts/decode_only.py 96 1 0 128 2>&1 | tail -1 | sed "s/^/[HEAD] /"
done
