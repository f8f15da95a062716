#!/bin/bash
# forward-only entry: tests + the inference probe of bench.py
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_step_plan.py tests/test_gpu_predict.py tests/test_gpu_train_entry.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/fwd_tests.log
timeout 900 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-cfg3 --no-split-probe > gpurun_out/fwd_bench.json 2> gpurun_out/fwd_bench.err
python - <<'PY' > gpurun_out/fwd_inf.txt
import json
l=[x for x in open('/root/repo/gpurun_out/fwd_bench.json') if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], d['ms_per_step'])
for k,v in d.get('inference',{}).items(): print(k, v)
PY
