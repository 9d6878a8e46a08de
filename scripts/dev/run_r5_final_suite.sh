#!/bin/bash
# round 5: the whole -m gpu suite + smoke() on the final tree (after the closing run one more prologue change went in)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 430 python -m pytest tests -m gpu -q > gpurun_out/r5_t_final.txt 2>&1; tail -3 gpurun_out/r5_t_final.txt
timeout 25 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r5_t_final.txt
