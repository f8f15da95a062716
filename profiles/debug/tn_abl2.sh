#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/tn2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/t_$1 -o t -- python3 $R/profiles/debug/gemm_p3_tn_time.py $2 > $O/$1.log 2>&1
  python3 $R/profiles/rocpd_summary.py $(ls $O/t_$1/*.db | head -1) $O/$1.csv > /dev/null
  echo "$1: kernel $(grep gemm_p3_tn_kernel $O/$1.csv | cut -d, -f4 | tr '\n' ' ')"
  rm -rf $O/t_$1
}
for sh in dw1 dw0; do
  run ${sh}_base $sh
  for v in 7 15 8; do ( export GTE_LIB_PATH=$R/profiles/micro/abl/lib_tnabl$v.so; run ${sh}_abl$v $sh ); done
done
