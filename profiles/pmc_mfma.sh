#!/bin/bash
# Matrix-pipe occupancy and effective clock of the train loop's GEMM kernels (the "power limit" claim of DESIGN.md section 5):
#   gpurun --timeout 900 -- 'bash profiles/pmc_mfma.sh r05'
# three separate --pmc passes (one counter each, kernel trace only; the program directly after `--`) over profiles/pmc_step.py,
# summarised per kernel by profiles/pmc_mfma_summary.py into gpurun_out/<tag>/pmc_mfma.txt
set -u
TAG=${1:-pmc}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $O/mfma/pmc_$C -o p --output-format csv -- python3 $R/profiles/pmc_step.py 12 > $O/pmc_mfma_$C.log 2>&1
done
cd $R
python3 profiles/pmc_mfma_summary.py $O/mfma > $O/pmc_mfma.txt
rm -rf $O/mfma
cat $O/pmc_mfma.txt
