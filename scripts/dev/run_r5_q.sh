#!/bin/bash
# round 5, GPU run Q: the two-process tests again (self-test with a patient first round), three times over
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3; do
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -s -k "peer_exchange or two_process or bench_tensor or row_split" > gpurun_out/r5q_t$i.txt 2>&1; tail -3 gpurun_out/r5q_t$i.txt
done
