#!/bin/bash
# round 5, GPU run N: row-split tensor-parallel decode with the exchanges folded into the producing launches (two processes, one GPU)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -s -k "peer_exchange or two_process or bench_tensor or row_split" > gpurun_out/r5n_t1.txt 2>&1; tail -15 gpurun_out/r5n_t1.txt
python scripts/decode_only.py 64 1 0 128 2>&1 | tail -2
for v in "folded:FL_X=1" "collectives:FL_TP_FOLD=0"; do
  n=${v%%:*}; e=${v#*:}
  env $e timeout 800 python scripts/dev/tp_decode_rehearsal.py 7B 2 /tmp/tpr_$n 64 2>&1 | grep -v "^$" | tail -4 | sed "s/^/[$n] /"
done
