#!/bin/bash
# round 5, GPU run P: the exchange tail in round 3's GEMV as well (65B widths: 7 -> 5 launches per layer), bench.py --gpus 2 rehearsal at 7B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -s -k "peer_exchange or two_process or bench_tensor or row_split" > gpurun_out/r5p_t1.txt 2>&1; tail -4 gpurun_out/r5p_t1.txt
timeout 1200 python -m pytest tests/test_exact_gpu.py tests/test_wide_models_gpu.py -m gpu -x -q > gpurun_out/r5p_t2.txt 2>&1; tail -3 gpurun_out/r5p_t2.txt
FL_P2P_MAX_COUNT=65536 FL_BENCH_DEVICE=0 FL_BENCH_BACKEND=gloo FL_BENCH_P2P_ONLY=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 \
  bench.py --gpus 2 --model 7B --n-batch 4 --steps 4 --warmup 1 --decode-steps 48 > gpurun_out/r5p_bench_tp2_7b.txt 2>&1
grep '^{' gpurun_out/r5p_bench_tp2_7b.txt > gpurun_out/r5p_bench_tp2_7b.json; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5p_bench_tp2_7b.json").read())
    print("bench tp2 7B:", {k: d.get(k) for k in ("value", "decode_tokens_per_s", "tp_decode", "tp_error", "tp_small_message_path")})
except Exception as e:
    print("bench tp2 7B failed:", e); print(open("gpurun_out/r5p_bench_tp2_7b.txt").read()[-3000:])
PY
for v in "folded:FL_X=1"; do
  n=${v%%:*}; e=${v#*:}
  env $e FL_LAYERS=4 timeout 800 python scripts/dev/tp_decode_rehearsal.py 65B 8 /tmp/tpr65_$n 64 2>&1 | grep "rank [01]" | sort | sed "s/^/[65B-width x 4 layers, G=8, $n] /"
  env $e timeout 800 python scripts/dev/tp_decode_rehearsal.py 13B 2 /tmp/tpr13_$n 64 2>&1 | grep "rank" | sort | sed "s/^/[13B, G=2, $n] /"
done
python scripts/decode_only.py 48 1 0 128 13B 2>&1 | tail -1
python scripts/decode_only.py 32 1 0 128 65B 2>&1 | tail -1
