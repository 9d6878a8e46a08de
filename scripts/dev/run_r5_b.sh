#!/bin/bash
# round 5, GPU run B: persistent / buffer-load form of the reference-order decode GEMV (tests + A/B), full-size parity, timeline, coexec6
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_exact_gpu.py tests/test_eval_ops_gpu.py -m gpu -x -q > gpurun_out/r5b_t1.txt 2>&1; tail -3 gpurun_out/r5b_t1.txt
FL_LLC_SLOTS=6 python -m pytest tests/test_exact_gpu.py tests/test_eval_ops_gpu.py -m gpu -x -q -k "gemv or pair or decode or model or forms" > gpurun_out/r5b_t2.txt 2>&1; tail -3 gpurun_out/r5b_t2.txt
for v in "default:" "nopersist:FL_LLC_NOPERSIST=1" ; do
  n=${v%%:*}; e=${v#*:}
  env $e python scripts/decode_only.py 64 1 0 128 2>&1 | grep decode | sed "s/^/[$n] /"
done
for v in "occ2:" "occ2-nopersist:FL_LLC_NOPERSIST=1" ; do
  n=${v%%:*}; e=${v#*:}
  env FASTLLAMA_HIP_LIB=gpurun_variants/libocc2.so $e python scripts/decode_only.py 64 1 0 128 2>&1 | grep decode | sed "s/^/[$n] /"
done
FASTLLAMA_HIP_LIB=gpurun_variants/libtl.so python scripts/dev/decode_timeline.py 7B 128 > gpurun_out/r5b_decode_timeline.md 2> gpurun_out/r5b_decode_timeline.err; tail -2 gpurun_out/r5b_decode_timeline.err; grep -v "^$" gpurun_out/r5b_decode_timeline.md | head -12
./scripts/ubench/coexec6 > gpurun_out/r5b_coexec6.txt 2>&1; cat gpurun_out/r5b_coexec6.txt
python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -s > gpurun_out/r5b_fullsize.txt 2>&1; tail -8 gpurun_out/r5b_fullsize.txt
