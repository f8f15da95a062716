#!/bin/bash
# the bench's long run against its timed region, several times on one box (is the difference the box or the order?)
R=$(cd $(dirname $0)/../.. && pwd)
F="--no-shapes --no-size-sweep --no-residency --no-cfg3 --no-cpu-baseline --no-inference --no-secondary --no-replay --no-split-probe"
for i in 1 2 3; do
  timeout 300 python3 $R/bench.py $F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full-order   long', round(d['long_run']['value']/1e6,2), 'timed', round(d['value']/1e6,2))"
  timeout 300 python3 $R/bench.py $F --no-gather-probe --val-graph 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-probes    long', round(d['long_run']['value']/1e6,2), 'timed', round(d['value']/1e6,2))"
done
LOOP_EPOCH=12 timeout 300 python3 $R/profiles/debug/loop_host_time.py
LOOP_EPOCH=240 timeout 300 python3 $R/profiles/debug/loop_host_time.py
nproc; uptime
