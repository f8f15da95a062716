#!/bin/bash
# Build a variant of libgte_hip.so with extra -D flags on ONE source: bash profiles/debug/build_variant.sh <name> <file.hip> "<flags>"
# -> profiles/micro/abl/lib_<name>.so   (use with GTE_LIB_PATH; ABL_DIR=<dir under profiles/micro> for another place: abl/ does not travel to the GPU box)
set -e
# (the other objects come from csrc/_build: bring them up to date first -- a stale object that disagrees on a header's structs links fine)
make -s -C $(cd $(dirname $0)/../.. && pwd)/gnn-tableextraction_amd/csrc
R=$(cd $(dirname $0)/../.. && pwd); C=$R/gnn-tableextraction_amd/csrc; O=$R/profiles/micro/${ABL_DIR:-abl}; mkdir -p $O
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-function $3 -c $C/$2 -o $O/$1.o
OBJS=$(ls $C/_build/*.o | grep -v "/$(basename $2 .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/lib_$1.so $OBJS $O/$1.o
echo built $O/lib_$1.so
