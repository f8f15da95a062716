#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, kernel trace only) of the train loop's kernels and of the cfg4
# aggregation: gpurun --timeout 900 -- 'bash profiles/pmc_refresh.sh r02pmc'
set -u
TAG=${1:-pmc}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GTE_GEMM_MODE=f32          # pass 1: the fp32 MFMA mode (the default of rounds 1-2)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace -d $O/step/pmc_$C -o p --output-format csv -- python3 $R/profiles/pmc_step.py 4 > $O/pmc_step_$C.log 2>&1
  timeout 200 rocprofv3 --pmc $C --kernel-trace -d $O/full/pmc_$C -o p --output-format csv -- python3 $R/profiles/pmc_probe.py spmm 4 > $O/pmc_$C.log 2>&1
done
unset GTE_GEMM_MODE               # pass 2: the default (split-bf16 arithmetic, planes GEMMs on P3 images)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace -d $O/step_split/pmc_$C -o p --output-format csv -- python3 $R/profiles/pmc_step.py 4 > $O/pmc_step_split_$C.log 2>&1
done
cd $R
python3 profiles/pmc_traffic_summary.py $O/full $O/pmc_per_kernel.json > $O/pmc_per_kernel.txt
python3 profiles/pmc_traffic_summary.py $O/step $O/pmc_per_kernel_step.json > $O/pmc_per_kernel_step.txt
python3 profiles/pmc_traffic_summary.py $O/step_split $O/pmc_per_kernel_step_split.json > $O/pmc_per_kernel_step_split.txt
NPS=$(grep -h "^PMC_STEP" $O/pmc_step_split_FETCH_SIZE.log | tail -1 | awk '{print $5}')
NST=$(grep -h "^PMC_STEP" $O/pmc_step_split_FETCH_SIZE.log | tail -1 | awk '{print $3}')       # the steps the driver really ran
python3 profiles/make_pmc_traffic.py $O/pmc_per_kernel_step.json $O/pmc_per_kernel.json $O/pmc_traffic.json $O/pmc_per_kernel_step_split.json ${NST:-4} ${NPS:-0} > /dev/null
rm -rf $O/full $O/step $O/step_split
cat $O/pmc_per_kernel_step.txt | head -20; cat $O/pmc_per_kernel.txt | head -5
