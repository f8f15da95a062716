#!/bin/bash
# SQ counters of every kernel of the train step (two passes over profiles/pmc_step.py): bash profiles/debug/pmc_step_sq.sh <tag>
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-pmc_sq}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_VALU --kernel-trace -d $O/a -o p --output-format csv -- python3 $R/profiles/pmc_step.py 3 > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-trace -d $O/b -o p --output-format csv -- python3 $R/profiles/pmc_step.py 3 > $O/b.log 2>&1
cd $R
python3 profiles/pmc_summary.py $O/a > $O/sq_a.txt
python3 profiles/pmc_summary.py $O/b > $O/sq_b.txt
rm -rf $O/a $O/b
wc -l $O/sq_a.txt $O/sq_b.txt
