#!/bin/bash
# kernel-only time of narrow_bwd_mfma_kernel per ablation build (NB_ABL bits, csrc/narrow_layer.hip) in the train loop:
#   ABL_DIR=abl_nb bash profiles/debug/build_variant.sh nb<bits> narrow_layer.hip "-DNB_ABL=<bits>";  bash profiles/debug/nb_abl.sh 1 2 4 ...
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/nb; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
STEP_ONLY="--no-cpu-baseline --no-gather-probe --no-secondary --no-cfg3 --no-replay --no-split-probe --no-inference --no-shapes --no-size-sweep --no-residency --no-uncached --no-dist-probe --no-kernel-timers --val-graph 0 --long-run-seconds 0.1"
for v in default "$@"; do
  ( [ $v != default ] && export GTE_LIB_PATH=$R/profiles/micro/abl_nb/lib_nb$v.so
    timeout 200 rocprofv3 --kernel-trace --stats -d $O/t_$v -o t -- python3 $R/bench.py $STEP_ONLY > $O/$v.log 2>&1 )
  python3 $R/profiles/rocpd_summary.py $(ls $O/t_$v/*.db | head -1) $O/$v.csv > /dev/null
  echo "abl $v: narrow_bwd $(grep narrow_bwd_mfma_kernel $O/$v.csv | cut -d, -f2,4,5) (calls, avg us, min us)"
  rm -rf $O/t_$v
done
