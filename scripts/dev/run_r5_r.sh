#!/bin/bash
# round 5, GPU run R: why is bench.py's two-rank decode (7.5 ms) slower than the rehearsal script's (2.9 ms)?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FL_NCTX=2048 FL_CHECK=0 timeout 800 python scripts/dev/tp_decode_rehearsal.py 7B 2 /tmp/tpr_a 64 2>&1 | grep "rank" | sort | sed "s/^/[n_ctx 2048] /"
for par in tp auto; do
FL_P2P_MAX_COUNT=65536 FL_BENCH_DEVICE=0 FL_BENCH_BACKEND=gloo FL_BENCH_P2P_ONLY=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 \
  bench.py --gpus 2 --model 7B --n-batch 4 --steps 4 --warmup 1 --decode-steps 48 --parallel $par --no-fast > gpurun_out/r5r_bench_$par.txt 2>&1
grep '^{' gpurun_out/r5r_bench_$par.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[bench --parallel $par]', d['decode_ms_per_token'], d['decode_device_resident']['ms_per_token'], d.get('tp_error'))"
done
