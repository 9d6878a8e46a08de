#!/bin/bash
# round 5, GPU run M: K.Q + soft_max in one launch (reference order), the prepare / warning test
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_exact_gpu.py -m gpu -x -q > gpurun_out/r5m_t1.txt 2>&1; tail -3 gpurun_out/r5m_t1.txt
python -m pytest tests/test_parity_7b_gpu.py tests/test_llama_api_gpu.py -m gpu -x -q -k "not floor" > gpurun_out/r5m_t2.txt 2>&1; tail -3 gpurun_out/r5m_t2.txt
for v in "fused:FL_X=1" "separate-soft_max:FL_XA_NOFUSE=1" "fused-again:FL_X=1" "separate-again:FL_XA_NOFUSE=1"; do
  n=${v%%:*}; e=${v#*:}
  env $e python scripts/prefill_only.py 8 512 0,512 2>&1 | grep prefill | sed "s/^/[$n] /"
done
