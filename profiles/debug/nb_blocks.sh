R=$PWD; O=$R/gpurun_out/nb; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
STEP_ONLY="--no-cpu-baseline --no-gather-probe --no-secondary --no-cfg3 --no-replay --no-split-probe --no-inference --no-shapes --no-size-sweep --no-residency --no-dist-probe --no-kernel-timers --val-graph 0 --long-run-seconds 0.1"
for sh in 831:256 13:218; do F=${sh%%:*}; H=${sh##*:}; for nb in 256 384 512; do
GTE_NARROW_BLOCKS=$nb timeout 200 rocprofv3 --kernel-trace --stats -d $O/t -o t -- python3 $R/bench.py --in-feats $F --hidden $H $STEP_ONLY > $O/x.log 2>&1
python3 $R/profiles/rocpd_summary.py $(ls $O/t/*.db | head -1) $O/x.csv > /dev/null
echo "F=$F H=$H blocks=$nb: narrow_bwd $(grep narrow_bwd_mfma_kernel $O/x.csv | cut -d, -f2,4,5)  fold $(grep gte_fold_batch $O/x.csv | cut -d, -f4)"
rm -rf $O/t; done; done
