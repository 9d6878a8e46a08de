#!/bin/bash
# per-kernel decode stats, round-4 LLC kernels vs round-3 kernels (FL_EXACT_R3=1), for one model: dec_ab.sh 65B
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; M=${1:-65B}
for v in llc r3; do
  if [ $v = r3 ]; then export FL_EXACT_R3=1; else unset FL_EXACT_R3; fi
  FL_NCTX=512 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dab_${M}_$v -o out --output-format csv -- python $R/scripts/decode_only.py 16 0 0 128 $M > $R/gpurun_out/prof_dab_${M}_$v.log 2>&1
  echo "== $M $v"; grep "decode" $R/gpurun_out/prof_dab_${M}_$v.log | tail -1
  python $R/scripts/dev/stats_summary.py $R/gpurun_out/prof_dab_${M}_$v | grep "gemv1\|decode_att" | head -8
done
