#!/bin/bash
# fold + Adam launch with the weight images: which images cost what (GTE_WIMG_ABL: 1 none, 2 untransposed only, 3 transposed only)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/wimg_abl
mkdir -p $O
STEP_ONLY="--no-cpu-baseline --no-gather-probe --no-secondary --no-cfg3 --no-replay --no-split-probe --no-inference --val-graph 0"
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3; do
  ( export GTE_WIMG_ABL=$v; timeout 300 rocprofv3 --kernel-trace --stats -d $O/t$v -o t -- python3 $R/bench.py --long-run-seconds 0.2 $STEP_ONLY > $O/t$v.log 2>&1 )
  python3 $R/profiles/rocpd_summary.py $(ls $O/t$v/*.db | head -1) $O/stats_$v.csv > /dev/null
  rm -rf $O/t$v
  echo "ABL=$v: $(grep fold_batch $O/stats_$v.csv | cut -c1-60)"
done
