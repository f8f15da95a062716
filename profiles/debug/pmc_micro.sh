#!/bin/bash
# SQ counters of the GEMM micro-benchmark variants (lone persistent workgroup per CU vs one tile per workgroup)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/pmc_micro; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --kernel-trace -d $O/a -o p --output-format csv -- $R/profiles/micro/gemm_persist 24576 512 ${1:-4096} > /dev/null 2>&1
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INSTS_VALU --kernel-trace -d $O/b -o p --output-format csv -- $R/profiles/micro/gemm_persist 24576 512 ${1:-4096} > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, os
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/pmc_micro")
for d in ("a", "b"):
    f = glob.glob(f"{root}/{d}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"]
        if "gemm_nt" not in k: continue
        key = k[k.index("gemm_nt"):][:40] + " grid=" + row["Grid_Size"]
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: f"{sorted(v)[len(v)//2]:.4g}" for c, v in cs.items()})
PY
rm -rf $O
