#!/bin/bash
# kernel stats of the train loop alone under environment variants, one trace each:
#   bash profiles/debug/trace_step_env.sh <out dir> "GTE_P3_ROWS=0" "GTE_P3_ROWS=1"
R=${GRAFT_REPO_ROOT:-$PWD}
O=$1; shift
mkdir -p $O
STEP_ONLY="--no-cpu-baseline --no-gather-probe --no-secondary --no-cfg3 --no-replay --no-split-probe --no-inference --val-graph 0 --no-shapes --no-size-sweep --no-residency"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  tag=$(echo $v | tr '= ' '__')
  ( export $v; timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$tag -o t -- python3 $R/bench.py --long-run-seconds 0.2 $STEP_ONLY > $O/trace_$tag.log 2>&1 )
  python3 $R/profiles/rocpd_summary.py $(ls $O/trace_$tag/*.db | head -1) $O/step_kernel_stats_$tag.csv > /dev/null
  rm -rf $O/trace_$tag
done
