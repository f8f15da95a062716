#!/bin/bash
# rocprofv3 kernel traces of the train loop on the reference's own run shapes (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash profiles/shapes_refresh.sh r04'
# writes gpurun_out/<tag>/shape_f<F0>_h<H>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py --in-feats F0 --hidden H, the
# train loop alone).  Every command is bounded by `timeout`.
set -u
TAG=${1:-x}
SHAPES=${2:-"831:1000 13:1000 13:218 363:149 363:139 831:96 781:100 313:157 63:206"}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
STEP_ONLY="--no-cpu-baseline --no-gather-probe --no-secondary --no-cfg3 --no-replay --no-split-probe --no-inference --no-shapes --no-size-sweep --no-residency --no-uncached --no-dist-probe --val-graph 0 --long-run-seconds 0.1 --resident-pages 400"
for sh in $SHAPES; do
  F=${sh%%:*}; H=${sh##*:}
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_${F}_${H} -o t -- python3 $R/bench.py --in-feats $F --hidden $H $STEP_ONLY > $O/shape_f${F}_h${H}.log 2>&1
  python3 $R/profiles/rocpd_summary.py $(ls $O/trace_${F}_${H}/*.db | head -1) $O/shape_f${F}_h${H}_kernel_stats.csv > /dev/null
  rm -rf $O/trace_${F}_${H}
done
cd $R
head -12 $O/shape_f831_h1000_kernel_stats.csv
