#!/bin/bash
# round 5, GPU run T: multi-pass rows in the 4 x 8 form (FL_LLC_MP): parity on the wide shapes, then decode A/B at 65B / 13B / 7B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FL_LLC_MP=2 timeout 1500 python -m pytest tests/test_exact_gpu.py tests/test_wide_models_gpu.py tests/test_full_size_gpu.py -m gpu -x -q > gpurun_out/r5t_t1.txt 2>&1; tail -3 gpurun_out/r5t_t1.txt
for mp in 0 1; do
  FL_LLC_MP=$mp python scripts/decode_only.py 32 1 0 128 65B 2>&1 | tail -1 | sed "s/^/[MP=$mp] /"
done
for mp in 0 1 2; do
  FL_LLC_MP=$mp python scripts/decode_only.py 48 1 0 128 13B 2>&1 | tail -1 | sed "s/^/[MP=$mp] /"
done
for mp in 0 1; do
  FL_LLC_MP=$mp python scripts/decode_only.py 64 1 0 128 7B 2>&1 | tail -1 | sed "s/^/[MP=$mp] /"
  FL_QTYPE=3 FL_LLC_MP=$mp python scripts/decode_only.py 64 1 0 128 7B 2>&1 | tail -1 | sed "s/^/[Q4_1 MP=$mp] /"
done
