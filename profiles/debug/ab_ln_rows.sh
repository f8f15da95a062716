timeout 900 python -m pytest tests/test_gemm_p3.py -q -m gpu -x -k "layernorm" > gpurun_out/t5.log 2>&1; tail -3 gpurun_out/t5.log
# (library switches live in the measurement build: both arms load it)
export GTE_LIB_PATH=${GTE_LIB_PATH:-$(cd $(dirname $0)/../.. && pwd)/gnn-tableextraction_amd/libgte_hip_measure.so}
STEP_ONLY="--no-cpu-baseline --no-gather-probe --no-secondary --no-cfg3 --no-replay --no-split-probe --no-inference --no-shapes --no-size-sweep --no-residency --no-uncached --no-dist-probe --no-kernel-timers --val-graph 0 --long-run-seconds 0.5"
for rep in 1 2; do for P in 25 50 75 100; do for v in 32 64 96 128; do
echo -n "rep $rep pages $P min_rows=$v: "; GTE_P3_LN_MIN_ROWS=$v timeout 300 python bench.py --pages $P $STEP_ONLY 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), d['ms_per_step'], 'long', round(d['long_run']['value']/1e6,2))"
done; done; done
