#!/bin/bash
# Refresh the measurements under profiles/ on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash profiles/refresh.sh f'
# writes gpurun_out/<tag>/: bench.json, kernel stats of the default command and of the step alone (rocprofv3
# --kernel-trace --stats), FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, kernel trace only), stream_bw yardstick.
set -u
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 $R/bench.py > $O/bench_trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace_step -o t -- python3 $R/bench.py --no-cpu-baseline --no-gather-probe --no-secondary > $O/bench_trace_step.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $O/full/pmc_$C -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-secondary > $O/pmc_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace -d $O/step/pmc_$C -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --no-gather-probe --no-secondary > $O/pmc_step_$C.log 2>&1
done
cd $R
python3 profiles/rocpd_summary.py $(ls $O/trace/*.db | head -1) $O/final_kernel_stats.csv > /dev/null
python3 profiles/rocpd_summary.py $(ls $O/trace_step/*.db | head -1) $O/step_kernel_stats.csv > /dev/null
python3 profiles/pmc_traffic_summary.py $O/full $O/pmc_per_kernel.json > $O/pmc_per_kernel.txt
python3 profiles/pmc_traffic_summary.py $O/step $O/pmc_per_kernel_step.json > $O/pmc_per_kernel_step.txt
[ -x profiles/micro/stream_bw ] && ./profiles/micro/stream_bw 2048 > $O/stream_bw.txt
rm -rf $O/trace $O/trace_step $O/full/*/*agent_info* $O/step/*/*agent_info*
tail -c 600 $O/bench.json
