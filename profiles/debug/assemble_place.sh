STEP_ONLY="--no-cpu-baseline --no-gather-probe --no-secondary --no-cfg3 --no-replay --no-split-probe --no-inference --val-graph 0"
for late in 1 0; do for rows in 0 48 192 768; do
  r=$( GTE_PIPE_LATE=$late GTE_ASSEMBLE_ROWS=$rows timeout 120 python bench.py $STEP_ONLY 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); print(round(d['value']/1e6,2), round(d['long_run']['value']/1e6,2), round(d['ms_per_step'],4))
")
  echo "late=$late rows=$rows : $r"
done; done
