#!/bin/bash
# (library switches live in the measurement build: both arms load it)
export GTE_LIB_PATH=${GTE_LIB_PATH:-$(cd $(dirname $0)/../.. && pwd)/gnn-tableextraction_amd/libgte_hip_measure.so}
# A/B of environment variants on the bench's train loop (headline + run shapes), interleaved on one box:
#   bash profiles/debug/ab_bench.sh "GTE_CACHE_AGG=1" "GTE_CACHE_AGG=0"
# prints value / ms_per_step / shapes per variant and round (bench_extras.json of each run is kept under gpurun_out/ab/)
mkdir -p gpurun_out/ab
FLAGS="--no-gather-probe --no-cfg3 --no-residency --no-uncached --no-size-sweep --no-inference --no-replay --val-graph 0 --no-cpu-baseline --no-split-probe --no-secondary --no-dist-probe ${AB_FLAGS}"
for round in 1 2; do
  i=0
  for v in "$@"; do
    i=$((i+1))
    env $v python bench.py $FLAGS > gpurun_out/ab/out_${i}_${round}.log 2> gpurun_out/ab/err_${i}_${round}.log
    cp bench_extras.json gpurun_out/ab/extras_${i}_${round}.json 2>/dev/null
    python - "$v" $round gpurun_out/ab/out_${i}_${round}.log <<'PY'
import json, sys
tag, rnd, path = sys.argv[1:4]
lines = [l for l in open(path) if l.startswith("{")]
if not lines:
    print(tag, rnd, "NO RECORD"); sys.exit(0)
d = json.loads(lines[-1])
sh = " ".join(f"{k}:{v[0]/1e6:.1f}" for k, v in d.get("shapes", {}).items())
print(f"{tag:28s} r{rnd} value {d['value']/1e6:.2f} M  {d['ms_per_step']:.4f} ms  long {d.get('long_run',{}).get('value',0)/1e6:.2f}  roof {d['roofline']['frac']:.3f} | {sh}")
PY
  done
done
