#!/bin/bash
# round 5, GPU run A: the full-size parity tests, the in-graph decode timeline, the bench line with its new fields
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -s > gpurun_out/r5a_fullsize.txt 2>&1; tail -12 gpurun_out/r5a_fullsize.txt
FASTLLAMA_HIP_LIB=gpurun_variants/libtl.so python scripts/dev/decode_timeline.py 7B 128 > gpurun_out/r5a_decode_timeline.md 2> gpurun_out/r5a_decode_timeline.err; tail -3 gpurun_out/r5a_decode_timeline.err; cat gpurun_out/r5a_decode_timeline.md
python bench.py > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err; tail -2 gpurun_out/r5a_bench.err; cut -c1-400 gpurun_out/r5a_bench.json
