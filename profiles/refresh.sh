#!/bin/bash
# Refresh the measurements under profiles/ on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash profiles/refresh.sh r02x'
# writes gpurun_out/<tag>/: bench.json (the default command), kernel stats of the default command and of the train loop alone
# (rocprofv3 --kernel-trace --stats; the >= 1 s long run shortened to 0.2 s under the profiler; the train loop also in the fp32 MFMA GEMM mode and at F0 = 13), FETCH_SIZE / WRITE_SIZE
# passes (separate --pmc runs, kernel trace only, over the short drivers profiles/pmc_step.py / pmc_probe.py: counter collection
# serialises every dispatch, bench.py under it runs for many minutes), copy yardsticks.  Every command is bounded by `timeout`.
set -u
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
STEP_ONLY="--no-cpu-baseline --no-gather-probe --no-secondary --no-cfg3 --no-replay --no-split-probe --no-inference --no-shapes --no-size-sweep --no-residency --no-uncached --no-dist-probe --val-graph 0"
timeout 400 python3 $R/bench.py > $O/bench.json 2> $O/bench.err        # the record line (what the driver parses) ...
cp $R/bench_extras.json $O/bench_extras.json 2> /dev/null                # ... and the full measurements behind it
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 $R/bench.py --long-run-seconds 0.2 --no-cpu-baseline > $O/bench_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_step -o t -- python3 $R/bench.py --long-run-seconds 0.2 $STEP_ONLY > $O/bench_trace_step.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_f32 -o t -- python3 $R/bench.py --gemm-mode f32 --long-run-seconds 0.2 $STEP_ONLY > $O/bench_trace_f32.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_13 -o t -- python3 $R/bench.py --in-feats 13 --long-run-seconds 0.2 $STEP_ONLY > $O/bench_trace_13.log 2>&1
cd $R
python3 profiles/rocpd_summary.py $(ls $O/trace_f32/*.db | head -1) $O/f32_step_kernel_stats.csv > /dev/null
python3 profiles/rocpd_summary.py $(ls $O/trace_13/*.db | head -1) $O/f13_step_kernel_stats.csv > /dev/null
timeout 200 python3 profiles/debug/gemm_p3_check.py > $O/gemm_p3_check.txt 2>&1
MLIB=$R/gnn-tableextraction_amd/libgte_hip_measure.so      # (forced tile configurations: a switch of the measurement build)
for c in 1 2 3 4 5 6; do GTE_LIB_PATH=$MLIB GTE_P3_NT_CFG=$c timeout 200 python3 profiles/debug/gemm_p3_nt_cfg.py 2>&1 | grep -v "amdgpu.ids\|bitwise" >> $O/gemm_p3_nt_cfg.txt; done
timeout 200 python3 profiles/debug/gemm_step_shapes.py > $O/gemm_step_shapes.txt 2>&1
timeout 300 python3 profiles/debug/gemm_split_check.py > $O/gemm_split_check.txt 2>&1
python3 profiles/rocpd_summary.py $(ls $O/trace/*.db | head -1) $O/final_kernel_stats.csv > /dev/null
python3 profiles/rocpd_summary.py $(ls $O/trace_step/*.db | head -1) $O/step_kernel_stats.csv > /dev/null
bash profiles/pmc_refresh.sh $TAG > /dev/null       # FETCH_SIZE / WRITE_SIZE passes over profiles/pmc_step.py and the cfg4 probe
bash profiles/pmc_mfma.sh $TAG > /dev/null          # round 5: matrix-pipe busy cycles / effective clock of the step's GEMM kernels
# round 4: the reference's run shapes, the validation-graph forward, planes GEMMs beyond one round of tiles, the host-resident set
bash profiles/shapes_refresh.sh $TAG > /dev/null
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_val -o t -- python3 $R/profiles/val_forward.py > $O/val_forward.log 2>&1)
python3 profiles/rocpd_summary.py $(ls $O/trace_val/*.db | head -1) $O/val_forward_kernel_stats.csv > /dev/null; rm -rf $O/trace_val
timeout 300 python3 profiles/gemm_p3_big_m.py 2> /dev/null > $O/gemm_p3_big_m.txt
for c in 2 4 5; do GTE_LIB_PATH=$MLIB GTE_P3_NT_CFG=$c timeout 300 python3 profiles/gemm_p3_big_m.py 2> /dev/null >> $O/gemm_p3_big_m.txt; done
timeout 300 python3 profiles/debug/residency_probe_alone.py 2> /dev/null > $O/residency_alone.txt
timeout 300 python3 profiles/residency_trace.py 3000 4 2.5 2> /dev/null | grep -v "^chunk" | head -12 > $O/residency_trace.txt
[ -x profiles/micro/stream_bw ] && timeout 60 ./profiles/micro/stream_bw 2048 > $O/stream_bw.txt
[ -x profiles/micro/copy_variants ] && timeout 60 ./profiles/micro/copy_variants 2048 > $O/copy_variants.txt
# round 6: the ablation table of the layer-0 forward kernels (binaries built by profiles/debug/l0_abl_build.sh travel with the tree)
[ -x profiles/micro/l0abl/l0_16 ] && bash profiles/debug/l0_abl_run.sh > /dev/null 2>&1 && cp gpurun_out/l0_fwd_ablation.txt $O/l0_fwd_ablation.txt
bash profiles/sequence_refresh.sh $TAG "831:256 831:96 13:218 831:1000 363:149" > /dev/null 2>&1
rm -rf $O/trace $O/trace_step $O/trace_f32 $O/trace_13
du -sh $O; tail -c 300 $O/bench.json
