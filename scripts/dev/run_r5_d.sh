#!/bin/bash
# round 5, GPU run D: scales-first / activation-first load order + early residual in the decode GEMV, pipelined V.P pieces, full-size parity
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_exact_gpu.py tests/test_eval_ops_gpu.py -m gpu -x -q > gpurun_out/r5d_t1.txt 2>&1; tail -3 gpurun_out/r5d_t1.txt
FL_LLC_SLOTS=6 python -m pytest tests/test_exact_gpu.py tests/test_eval_ops_gpu.py -m gpu -x -q -k "gemv or pair or decode or model or forms" > gpurun_out/r5d_t2.txt 2>&1; tail -3 gpurun_out/r5d_t2.txt
for v in "default:FL_X=1" "late:FASTLLAMA_HIP_LIB=gpurun_variants/liblate.so" "default-again:FL_X=1" "persist:FL_LLC_PERSIST=1"; do
  n=${v%%:*}; e=${v#*:}
  env $e python scripts/decode_only.py 64 1 0 128 2>&1 | grep decode | sed "s/^/[$n] /"
done
python scripts/dev/prefill_deep.py 7B 0,512,1024,1536 2>&1 | grep n_past
python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -s > gpurun_out/r5d_fullsize.txt 2>&1; tail -8 gpurun_out/r5d_fullsize.txt
