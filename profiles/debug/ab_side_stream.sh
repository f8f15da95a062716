STEP_ONLY="--no-cpu-baseline --no-gather-probe --no-secondary --no-cfg3 --no-replay --no-split-probe --no-inference --no-shapes --no-size-sweep --no-residency --no-uncached --no-dist-probe --no-kernel-timers --val-graph 0 --long-run-seconds 0.5"
# (library switches live in the measurement build: both arms load it)
export GTE_LIB_PATH=${GTE_LIB_PATH:-$(cd $(dirname $0)/../.. && pwd)/gnn-tableextraction_amd/libgte_hip_measure.so}
for rep in 1 2; do for sh in 831:256 13:218 831:96 13:256; do F=${sh%%:*}; H=${sh##*:}; for v in 1 0; do
echo -n "rep $rep F=$F H=$H side=$v: "; GTE_PIPE_SIDE=$v timeout 300 python bench.py --in-feats $F --hidden $H $STEP_ONLY 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), d['ms_per_step'], 'long', d['long_run']['value']/1e6)"
done; done; done
