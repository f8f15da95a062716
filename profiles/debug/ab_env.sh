#!/bin/bash
# A/B of environment switches on ONE box, interleaved: bash profiles/debug/ab_env.sh "GTE_C_STEP=0" "GTE_C_STEP=1" ...
# prints value (M nodes/s), long_run (M nodes/s) and ms/step of the train loop alone per variant and round.
STEP_ONLY="--no-cpu-baseline --no-gather-probe --no-secondary --no-cfg3 --no-replay --no-split-probe --no-inference --val-graph 0 --no-shapes --no-size-sweep --no-residency"
ROUNDS=${ROUNDS:-2}
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    out=$(env $v timeout 150 python bench.py $STEP_ONLY $EXTRA 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); print(round(d['value']/1e6,2), round(d['long_run']['value']/1e6,2), round(d['ms_per_step'],4))
")
    echo "round $r [$v]: $out"
  done
done
