#!/bin/bash
# SQ counters of the step's GEMM launches: bash profiles/debug/pmc_gemm.sh <outdir>   (on the GPU box; export GTE_GEMM_MODE=split
# first for the split kernels)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-pmc_gemm}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for C in l0fwd l1fwd l1dx l0dw; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --kernel-trace -d $O/a_$C -o p --output-format csv -- python3 $R/profiles/debug/gemm_case.py $C 6 > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD --kernel-trace -d $O/b_$C -o p --output-format csv -- python3 $R/profiles/debug/gemm_case.py $C 6 > /dev/null 2>&1
done
cd $R; python3 profiles/debug/pmc_gemm_summary.py $O
