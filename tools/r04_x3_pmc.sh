#!/bin/bash
# Round-4 SQ / HBM counters for the bf16x3 kernels built this round: the Linear GEMM on the LDS-DMA kernel (both operands split in
# registers / weights as pre-split words) at the step's QKV shape, and the conv weight gradient on (hi, lo) bf16 images at 96^3.
# Run on the GPU box from the repo root; summaries land in gpurun_out/r04_x3_pmc/.
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O=$ROOT/gpurun_out/r04_x3_pmc
mkdir -p $O
PREC=bf16x3 KERNEL=gemm FILTER=gemm_bf16_kernel KARGS="--m 432 --n 2304 --k 768" FAMILY="bf16x3 Linear GEMMs (X3 = 1: both operands split in registers)" \
  SHAPE="qkv 432x2304x768, fp32 storage" bash $ROOT/tools/pmc_conv3.sh $O/gemm_x3 > $O/gemm_x3.log 2>&1
cp $O/gemm_x3/pmc_summary.json $O/r04_pmc_gemm_x3_qkv_432rows.json
PREC=bf16x3 KERNEL=gemm_x3w FILTER=gemm_bf16_kernel KARGS="--m 432 --n 2304 --k 768" FAMILY="bf16x3 Linear GEMMs (X3 = 2: weights as pre-split words)" \
  SHAPE="qkv 432x2304x768, fp32 activations, word-shadow weights" bash $ROOT/tools/pmc_conv3.sh $O/gemm_x3w > $O/gemm_x3w.log 2>&1
cp $O/gemm_x3w/pmc_summary.json $O/r04_pmc_gemm_x3words_qkv_432rows.json
PREC=bf16x3 KERNEL=conv3_wgrad3 FILTER=conv3_wgrad_x3_kernel KARGS="--cin 32 --cout 16 --size 96 --batch 2" FAMILY="bf16x3 3x3x3 conv weight-grad" \
  SHAPE="dw[16,32,27] + dw3[16,32] @ 96^3 x 2, fp32 storage" bash $ROOT/tools/pmc_conv3.sh $O/wgrad_x3 > $O/wgrad_x3.log 2>&1
cp $O/wgrad_x3/pmc_summary.json $O/r04_pmc_conv3_wgrad_x3_32to16_96cube.json
echo "x3 pmc done"
