#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FASTLLAMA_HIP_LIB=gpurun_variants/libtl.so python scripts/dev/decode_timeline.py 7B 128 > gpurun_out/r5i_decode_timeline.md 2> gpurun_out/r5i_decode_timeline.err; tail -2 gpurun_out/r5i_decode_timeline.err; grep -v "^$" gpurun_out/r5i_decode_timeline.md | head -12
