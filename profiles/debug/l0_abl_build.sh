#!/bin/bash
# Ablation builds of the layer-0 forward micro-benchmark (profiles/micro/l0_fwd_abl.hip) -> profiles/micro/l0abl/ (travels to the GPU box)
# bash profiles/debug/l0_abl_build.sh [abl bits ...]
R=$(cd $(dirname $0)/../.. && pwd); O=$R/profiles/micro/l0abl; mkdir -p $O
H="/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-unused-function -Wno-unused-value -Wno-unused-result"
[ -f $O/gte_core.o ] || $H -c $R/gnn-tableextraction_amd/csrc/gte_core.hip -o $O/gte_core.o
for a in ${@:-0 16 17 20 24 48 80 144 29}; do
  ( $H ${EXTRA} -DP3_ABL=$a -c $R/profiles/micro/l0_fwd_abl.hip -o $O/l0_$a.o && $H $O/l0_$a.o $O/gte_core.o -o $O/l0_${TAG}$a && rm $O/l0_$a.o ) &
done
wait
ls -la $O
