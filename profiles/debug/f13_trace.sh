#!/bin/bash
# kernel stats of the F0 = 13 train loop + its nodes/s: bash profiles/debug/f13_trace.sh <tag>
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; T=${1:-f13}
STEP_ONLY="--no-cpu-baseline --no-gather-probe --no-secondary --no-cfg3 --no-split-probe --no-inference --val-graph 0"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/t_$T -o t -- python3 $R/bench.py --in-feats 13 --long-run-seconds 0.2 $STEP_ONLY --no-replay > /dev/null 2>&1
python3 $R/profiles/rocpd_summary.py $(ls $O/t_$T/*.db | head -1) $O/$T.csv > /dev/null
rm -rf $O/t_$T
head -${2:-14} $O/$T.csv | cut -c1-110
timeout 200 python3 $R/bench.py --in-feats 13 $STEP_ONLY 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('F0=13 loop', round(d['value']/1e6, 2), 'M nodes/s', round(d['ms_per_step'], 4), 'ms; long', round(d['long_run']['value']/1e6, 2), 'replay', round(d['replay']['value']/1e6, 2) if d.get('replay') else None)"
