#!/bin/bash
# One steady-state step of the train loop in launch order (profiles/rocpd_sequence.py) for a list of shapes:
#   gpurun --timeout 900 -- 'bash profiles/sequence_refresh.sh r05 "831:256 831:1000 13:218 831:96"'
# writes gpurun_out/<tag>/sequence_f<F0>_h<H>.txt.  Every command is bounded by `timeout`.
set -u
TAG=${1:-x}
SHAPES=${2:-"831:256 831:1000 13:218 831:96 363:149"}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
STEP_ONLY="--no-cpu-baseline --no-gather-probe --no-secondary --no-cfg3 --no-replay --no-split-probe --no-inference --no-shapes --no-size-sweep --no-residency --no-uncached --no-dist-probe --no-kernel-timers --val-graph 0 --long-run-seconds 0.1 --resident-pages 400"
for sh in $SHAPES; do
  F=${sh%%:*}; H=${sh##*:}
  timeout 300 rocprofv3 --kernel-trace -d $O/seq_${F}_${H} -o t -- python3 $R/bench.py --in-feats $F --hidden $H $STEP_ONLY > $O/seq_f${F}_h${H}.log 2>&1
  python3 $R/profiles/rocpd_sequence.py $(ls $O/seq_${F}_${H}/*.db | head -1) $O/sequence_f${F}_h${H}.txt > /dev/null 2>> $O/seq_f${F}_h${H}.log
  rm -rf $O/seq_${F}_${H}
done
cd $R
cat $O/sequence_f*.txt
