#!/bin/bash
# timings of the in-tree H16 exact GEMM and of every library under gpurun_variants/ on two shapes (ablation builds: results may be wrong)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== in-tree"; XH_QUICK=1 python scripts/dev/xh_check.py --perf-only
for f in gpurun_variants/lib_*.so; do echo "== $f"; FASTLLAMA_HIP_LIB=$PWD/$f XH_QUICK=1 python scripts/dev/xh_check.py --perf-only; done
} > gpurun_out/xh_variants.txt 2>&1
grep -v amdgpu.ids gpurun_out/xh_variants.txt
