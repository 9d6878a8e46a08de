#!/bin/bash
# round 5, GPU run Z3: batched rest loads only for long rests: 13B against the commit before, 65B once, wide-shape parity
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do
  FASTLLAMA_HIP_LIB=$PWD/gpurun_variants/libprev.so python scripts/dev/decode_ab_lib.py 48 1 0 128 13B 2>&1 | tail -1 | sed "s/^/[before] /"
  python scripts/decode_only.py 48 1 0 128 13B 2>&1 | tail -1 | sed "s/^/[HEAD] /"
done
python scripts/decode_only.py 32 1 0 128 65B 2>&1 | tail -1 | sed "s/^/[HEAD] /"
timeout 1500 python -m pytest tests/test_exact_gpu.py tests/test_wide_models_gpu.py tests/test_full_size_gpu.py -m gpu -x -q > gpurun_out/r5z3_t1.txt 2>&1; tail -2 gpurun_out/r5z3_t1.txt
