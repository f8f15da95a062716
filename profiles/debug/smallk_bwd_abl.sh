#!/bin/bash
# kernel-only time of sage_smallk_bwd_kernel per ablation build (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/skb; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in default "$@"; do
  ( [ $v != default ] && export GTE_LIB_PATH=$R/profiles/micro/abl/lib_skb$v.so
    timeout 120 rocprofv3 --kernel-trace --stats -d $O/t_$v -o t -- python3 $R/profiles/debug/smallk_bwd_time.py > $O/$v.log 2>&1 )
  python3 $R/profiles/rocpd_summary.py $(ls $O/t_$v/*.db | head -1) $O/$v.csv > /dev/null
  echo "abl $v: kernel $(grep sage_smallk_bwd_kernel $O/$v.csv | cut -d, -f4) us; folds $(grep smallk_fold $O/$v.csv | cut -d, -f2,4)"
  rm -rf $O/t_$v
done
