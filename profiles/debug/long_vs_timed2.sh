#!/bin/bash
# which part of the default bench command slows its long run (37 M against 44 M in the timed region)?
R=$(cd $(dirname $0)/../.. && pwd)
run() { timeout 300 python3 $R/bench.py --no-cpu-baseline $2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: long', round(d['long_run']['value']/1e6,2), 'timed', round(d['value']/1e6,2))"; }
run "A default           " ""
run "B no val graph      " "--val-graph 0"
run "C no gather probe   " "--no-gather-probe"
run "D no extra page sets" "--no-shapes --no-size-sweep --no-residency"
run "E no val, no gather " "--val-graph 0 --no-gather-probe"
run "A default           " ""
