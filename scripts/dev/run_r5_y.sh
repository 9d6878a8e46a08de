#!/bin/bash
# round 5, GPU run Y: the tail as its own instantiation of the GEMV: tensor-parallel tests, the peer that never arrives, wide shapes, rehearsal
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -s -k "two_process or bench_tensor or row_split" > gpurun_out/r5y_t1.txt 2>&1; tail -5 gpurun_out/r5y_t1.txt
timeout 1200 python -m pytest tests/test_exact_gpu.py tests/test_wide_models_gpu.py tests/test_eval_ops_gpu.py -m gpu -x -q > gpurun_out/r5y_t2.txt 2>&1; tail -2 gpurun_out/r5y_t2.txt
timeout 800 python scripts/dev/tp_decode_rehearsal.py 7B 2 /tmp/tpr7 64 2>&1 | grep "rank" | sort | sed "s/^/[7B, G=2] /"
FL_LAYERS=4 timeout 800 python scripts/dev/tp_decode_rehearsal.py 65B 8 /tmp/tpr65 64 2>&1 | grep "rank [01]" | sort | sed "s/^/[65B-width x 4 layers, G=8] /"
