#!/bin/bash
# the model-level GPU tests under every engine switch (one variant per line): bash profiles/debug/variant_tests.sh
FILES="tests/test_gpu_parity.py tests/test_gpu_step_plan.py tests/test_gpu_train_entry.py tests/test_gpu_predict.py tests/test_gpu_distributed.py"
for v in "GTE_C_STEP=0" "GTE_P3_ROWS=0" "GTE_FUSE_ADAM=0" "GTE_TAIL_SPLIT=0" "GTE_FUSED_HEAD=0" "GTE_TRANSFORM_FIRST=0" "GTE_PIPE_LATE=0" \
         "GTE_OVERLAP_DW=1" "GTE_FUSE_LN_BELOW=1" "GTE_STEP_GRAPH=1" "GTE_FUSE_LN_FWD=0" "GTE_SMALLK=0" "GTE_C_STEP=0 GTE_FUSE_LN_DX=0" "GTE_WIMG_IN_FOLD=0"; do
  out=$(env $v timeout 900 python -m pytest $FILES -q -m gpu -x 2>&1 | tail -3 | tr '\n' ' ')
  echo "[$v] $out"
done
