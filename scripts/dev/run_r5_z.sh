#!/bin/bash
# round 5, GPU run Z: the Q8_0 prologue's rest loads four iterations at a time (K > 8192 on 256 threads): parity, decode at 65B / 13B / 7B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_exact_gpu.py tests/test_wide_models_gpu.py tests/test_full_size_gpu.py tests/test_eval_ops_gpu.py -m gpu -x -q > gpurun_out/r5z_t1.txt 2>&1; tail -2 gpurun_out/r5z_t1.txt
python scripts/decode_only.py 32 1 0 128 65B 2>&1 | tail -1
python scripts/decode_only.py 48 1 0 128 13B 2>&1 | tail -1
python scripts/decode_only.py 64 1 0 128 7B 2>&1 | tail -1
FL_QTYPE=3 python scripts/decode_only.py 64 1 0 128 7B 2>&1 | tail -1
