#!/bin/bash
# (library switches live in the measurement build: both arms load it)
export GTE_LIB_PATH=${GTE_LIB_PATH:-$(cd $(dirname $0)/../.. && pwd)/gnn-tableextraction_amd/libgte_hip_measure.so}
# A/B of one environment switch on the train loop alone (un-profiled), interleaved:  bash profiles/debug/ab_env.sh VAR "v1 v2 ..." "F:H F:H ..." [reps]
VAR=$1; VALS=$2; SHAPES=${3:-"831:256"}; REPS=${4:-2}
STEP_ONLY="--no-cpu-baseline --no-gather-probe --no-secondary --no-cfg3 --no-replay --no-split-probe --no-inference --no-shapes --no-size-sweep --no-residency --no-uncached --no-dist-probe --no-kernel-timers --val-graph 0 --long-run-seconds 0.5"
for rep in $(seq 1 $REPS); do for sh in $SHAPES; do F=${sh%%:*}; H=${sh##*:}; for v in $VALS; do
echo -n "rep $rep F=$F H=$H $VAR=$v: "; env $VAR=$v timeout 300 python bench.py --in-feats $F --hidden $H $STEP_ONLY 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), d['ms_per_step'], 'long', round(d['long_run']['value']/1e6,2))"
done; done; done
