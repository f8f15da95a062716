R=$PWD; O=$R/gpurun_out/seqp; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
STEP_ONLY="--no-cpu-baseline --no-gather-probe --no-secondary --no-cfg3 --no-replay --no-split-probe --no-inference --no-shapes --no-size-sweep --no-residency --no-dist-probe --no-kernel-timers --val-graph 0 --long-run-seconds 0.1 --resident-pages 400"
for P in 50 25; do
timeout 300 rocprofv3 --kernel-trace -d $O/t$P -o t -- python3 $R/bench.py --pages $P $STEP_ONLY > $O/p$P.log 2>&1
python3 $R/profiles/rocpd_sequence.py $(ls $O/t$P/*.db | head -1) $O/sequence_pages$P.txt > /dev/null; rm -rf $O/t$P
done
cd $R; cat $O/sequence_pages*.txt
for P in 100 50 25; do echo -n "pages $P: "; timeout 300 python bench.py --pages $P $STEP_ONLY 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), d['ms_per_step'], 'long', round(d['long_run']['value']/1e6,2))"; done
