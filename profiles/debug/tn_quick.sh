#!/bin/bash
# kernel-only time of the two dW GEMMs (rocprofv3 kernel trace), default build: bash profiles/debug/tn_quick.sh <tag> [zero]
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-tnq}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for sh in dw1 dw0; do
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/t_$sh -o t -- python3 $R/profiles/debug/gemm_p3_tn_time.py $sh $2 > $O/$sh.log 2>&1
  python3 $R/profiles/rocpd_summary.py $(ls $O/t_$sh/*.db | head -1) $O/$sh.csv > /dev/null
  echo "$sh: $(grep 'TF fp32' $O/$sh.log | tail -1) | kernel: $(grep gemm_p3_tn_kernel $O/$sh.csv | cut -d, -f2,4 | tr '\n' ' ')  fold: $(grep p3_fold $O/$sh.csv | cut -d, -f4)"
  rm -rf $O/t_$sh
done
