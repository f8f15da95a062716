#!/bin/bash
# TN (dW) planes kernel study: kernel-only times (rocprofv3 kernel trace) per shape, split count, ablation build, operand values.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-tn}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # tag, shape, extra args ; env from caller
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/t_$1 -o t -- python3 $R/profiles/debug/gemm_p3_tn_time.py $2 $3 > $O/$1.log 2>&1
  python3 $R/profiles/rocpd_summary.py $(ls $O/t_$1/*.db | head -1) $O/$1.csv > /dev/null
  echo "$1: $(grep 'TF fp32' $O/$1.log | tail -1) | kernel: $(grep gemm_p3_tn_kernel $O/$1.csv | cut -d, -f2,4 | tr '\n' ' ')  fold: $(grep p3_fold $O/$1.csv | cut -d, -f4)"
  rm -rf $O/t_$1
}
for sh in dw1 dw0; do
  run ${sh}_base $sh
  run ${sh}_zero $sh zero
  for sp in 8 16 32 128; do ( export GTE_P3_TN_SPLITS=$sp; run ${sh}_sp$sp $sh ); done
  for v in 1 2 4 6 7; do
    [ -f $R/profiles/micro/abl/lib_tnabl$v.so ] && ( export GTE_LIB_PATH=$R/profiles/micro/abl/lib_tnabl$v.so; run ${sh}_abl$v $sh )
  done
done
