#!/bin/bash
# runs the builds of profiles/debug/l0_abl_build.sh on the GPU box; output -> gpurun_out/l0_fwd_ablation.txt
cd $(dirname $0)/../..; O=profiles/micro/l0abl; mkdir -p gpurun_out
{
echo "# layer-0 forward [x | cached ahn] W^T, 24 400 x 256 x 1 662, LayerNorm + ReLU epilogue, gemm_p3_nt_lw_kernel<..., LNB = 4> (profiles/micro/l0_fwd_abl.hip)"
echo "# abl bits: 1 no LDS-DMA in the loop, 4 fragments read once, 8 no MFMA, 16 in-kernel clock stamps, 32 no epilogue, 64 / 128 the A / B requests move nothing"
for a in ${ABLS:-0 16 17 20 24 29 48 80 144}; do
  [ -x $O/l0_$a ] && timeout 120 $O/l0_$a gemm
done
[ -x $O/l0_16 ] && ABL_ZERO=1 timeout 120 $O/l0_16 gemm
[ -x $O/l0_16 ] && ABL_ROWS=128 timeout 120 $O/l0_16 gemm
[ -x $O/l0_16 ] && ABL_ROWS=128 timeout 120 $O/l0_16 gemm 104
[ -x $O/l0_17 ] && ABL_ROWS=128 timeout 120 $O/l0_17 gemm 104
echo "# the block-major-weights kernel (gemm_p3_nt_sq_kernel) on the same problem; first: element by element against the kernel above"
[ -x $O/l0_16 ] && ABL_REF=1 ABL_SQ=1 timeout 120 $O/l0_16 gemm
for a in 17 24 48; do [ -x $O/l0_$a ] && ABL_SQ=1 timeout 120 $O/l0_$a gemm; done
[ -x $O/l0_16 ] && ABL_SQ=1 ABL_ZERO=1 timeout 120 $O/l0_16 gemm
[ -x $O/l0_16 ] && ABL_SQ=1 ABL_ROWS=128 timeout 120 $O/l0_16 gemm 104
[ -x $O/l0_16 ] && ABL_BBLK=1 timeout 120 $O/l0_16 gemm
echo
timeout 300 $O/l0_0 dma
} 2>&1 | tee gpurun_out/l0_fwd_ablation.txt
