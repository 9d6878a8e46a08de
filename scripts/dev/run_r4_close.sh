#!/bin/bash
# closing run of round 4 on one box: the whole -m gpu suite, smoke(), the bench line, kernel stats, PMC traffic, the PCIe-inclusive rate
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/t_close.txt 2>&1; tail -3 gpurun_out/t_close.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python scripts/dev/pcie_inclusive.py 2>&1 | tail -1 | tee gpurun_out/pcie_inclusive.txt
bash scripts/dev/run_r4_final2.sh
bash scripts/dev/run_r4_final.sh > gpurun_out/run_r4_final.log 2>&1
cut -c1-200 gpurun_out/bench_r4_final.json
