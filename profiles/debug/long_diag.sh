#!/bin/bash
# the default bench command several times with per-call timing of its long run (GTE_BENCH_LONG_DIAG=1): where does an occasional
# 37 M long run (against 44 M in the timed region) lose its time?
R=$(cd $(dirname $0)/../.. && pwd)
for i in 1 2 3 4 5 6; do
  GTE_BENCH_LONG_DIAG=1 timeout 300 python3 $R/bench.py --no-cpu-baseline --no-shapes --no-size-sweep --no-residency --no-cfg3 --no-inference --no-secondary --no-replay --no-split-probe 2> /tmp/diag.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i: long', round(d['long_run']['value']/1e6,2), 'timed', round(d['value']/1e6,2))"
  grep "long run, per" /tmp/diag.err
done
