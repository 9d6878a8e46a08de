#!/bin/bash
# round 5, GPU run S: producers push their rows to the peers themselves (tp_put), relaxed polls: fold tests + rehearsals
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -s -k "two_process or bench_tensor or row_split" > gpurun_out/r5s_t1.txt 2>&1; tail -3 gpurun_out/r5s_t1.txt
for i in 1 2; do
FL_CHECK=$((2-i)) timeout 800 python scripts/dev/tp_decode_rehearsal.py 7B 2 /tmp/tpr_$i 64 2>&1 | grep "rank" | sort | sed "s/^/[7B G=2 pass $i] /"
done
FL_LAYERS=4 timeout 800 python scripts/dev/tp_decode_rehearsal.py 65B 8 /tmp/tpr65 64 2>&1 | grep "rank [01]" | sort | sed "s/^/[65B-width x 4 layers, G=8] /"
timeout 800 python scripts/dev/tp_decode_rehearsal.py 13B 2 /tmp/tpr13 64 2>&1 | grep "rank" | sort | sed "s/^/[13B, G=2] /"
python scripts/decode_only.py 64 1 0 128 2>&1 | tail -1
