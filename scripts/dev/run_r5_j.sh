#!/bin/bash
# round 5, GPU run J: Q4_1 decode -- m_w broadcast in the chain; the 4 x 8 form against the 8 x 4 form
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_exact_gpu.py tests/test_eval_ops_gpu.py -m gpu -x -q -k "gemv or pair or decode or model or forms" > gpurun_out/r5j_t1.txt 2>&1; tail -2 gpurun_out/r5j_t1.txt
FL_Q41_48=1 python -m pytest tests/test_exact_gpu.py tests/test_eval_ops_gpu.py -m gpu -x -q -k "gemv or pair or decode or model or forms" > gpurun_out/r5j_t2.txt 2>&1; tail -2 gpurun_out/r5j_t2.txt
for v in "q41-8x4:FL_QTYPE=3" "q41-4x8:FL_QTYPE=3 FL_Q41_48=1" "q41-8x4-again:FL_QTYPE=3" "q41-4x8-again:FL_QTYPE=3 FL_Q41_48=1" "q40:FL_X=1"; do
  n=${v%%:*}; e=${v#*:}
  env $e python scripts/decode_only.py 64 1 0 128 2>&1 | grep decode | sed "s/^/[$n] /"
done
