#!/bin/bash
# (library switches live in the measurement build: both arms load it)
export GTE_LIB_PATH=${GTE_LIB_PATH:-$(cd $(dirname $0)/../.. && pwd)/gnn-tableextraction_amd/libgte_hip_measure.so}
# the run shapes of the reference under environment variants, interleaved on one box:
#   bash profiles/debug/ab_shapes.sh "GTE_FUSE_LN_DX=1" "GTE_FUSE_LN_DX=0"      -> M nodes/s and ms/step per shape
F="--no-cpu-baseline --no-gather-probe --no-cfg3 --no-replay --no-split-probe --no-inference --val-graph 0 --no-size-sweep --no-residency --no-uncached --long-run-seconds 0"
ROUNDS=${ROUNDS:-2}
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    env $v timeout 300 python bench.py $F 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
out = ['headline %.2f' % (d['value'] / 1e6)]
for k, s in d['shapes'].items():
    if isinstance(s, dict) and 'value' in s: out.append('%s %.2f (%.4f ms)' % (k, s['value'] / 1e6, s['ms_per_step']))
print('round $r [$v]: ' + '  '.join(out))
"
  done
done
