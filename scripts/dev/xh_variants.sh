#!/bin/bash
# time the H16 exact GEMM of every library under gpurun_variants/ (and the in-tree one) on two shapes; then SQ counters of the in-tree one
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== in-tree"; XH_QUICK=1 python scripts/dev/xh_check.py --perf-only
for f in gpurun_variants/lib_*.so; do echo "== $f"; FASTLLAMA_HIP_LIB=$PWD/$f XH_QUICK=1 python scripts/dev/xh_check.py --perf-only; done
} > gpurun_out/xh_variants.txt 2>&1
KPAT=gemm_q4_exact_h16 bash scripts/dev/pmc_gx.sh xh 2 12288 4096 512 3 5 > gpurun_out/xh_pmc.txt 2>&1
cat gpurun_out/xh_variants.txt | grep -v amdgpu.ids; cat gpurun_out/xh_pmc.txt
