R=$PWD; O=$R/gpurun_out/seqdp; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
STEP_ONLY="--no-cpu-baseline --no-gather-probe --no-secondary --no-cfg3 --no-replay --no-split-probe --no-inference --no-shapes --no-size-sweep --no-residency --no-dist-probe --no-kernel-timers --val-graph 0 --long-run-seconds 0.1 --resident-pages 400"
export GTE_BENCH_FORCE_DIST=1
timeout 300 rocprofv3 --kernel-trace -d $O/t -o t -- python3 $R/bench.py $STEP_ONLY > $O/log.txt 2>&1
python3 $R/profiles/rocpd_sequence.py $(ls $O/t/*.db | head -1) $O/sequence_dp1.txt; rm -rf $O/t
