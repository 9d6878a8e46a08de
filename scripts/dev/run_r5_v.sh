#!/bin/bash
# round 5, GPU run V: multi-pass rows, quads per wave and pass 8 / 7 / 6 (two vs three workgroups per CU): decode at 65B / 13B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for q in 8 7 6; do
  FL_LLC_MPQ=$q python scripts/decode_only.py 32 1 0 128 65B 2>&1 | tail -1 | sed "s/^/[MPQ=$q] /"
done
for q in 8 7 6; do
  FL_LLC_MPQ=$q python scripts/decode_only.py 48 1 0 128 13B 2>&1 | tail -1 | sed "s/^/[MPQ=$q] /"
done
FL_LLC_MP=2 timeout 1500 python -m pytest tests/test_exact_gpu.py tests/test_wide_models_gpu.py -m gpu -x -q > gpurun_out/r5v_t1.txt 2>&1; tail -2 gpurun_out/r5v_t1.txt
FL_LLC_MP=2 FL_LLC_MPQ=6 timeout 1500 python -m pytest tests/test_exact_gpu.py tests/test_wide_models_gpu.py -m gpu -x -q > gpurun_out/r5v_t2.txt 2>&1; tail -2 gpurun_out/r5v_t2.txt
