#!/bin/bash
# Round-4 SQ / HBM counters for the two conv kernels VERDICT r03 names (the fused residual-block input gradient at 96^3 and the
# weight gradient with the 1x1x1 branch at 96^3).  Run on the GPU box from the repo root; summaries land in gpurun_out/r04_conv_pmc/.
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O=$ROOT/gpurun_out/r04_conv_pmc
mkdir -p $O
KERNEL=conv3_dgrad_fused FILTER=conv3_fwd_pipe_kernel KARGS="--cin 32 --cout 16 --size 96 --batch 2" FAMILY="3x3x3 conv fwd + data-grad" \
  SHAPE="dx[32] = conv3^T(dc1[16]) + conv1^T(dc3[16]) @ 96^3 x 2" bash $ROOT/tools/pmc_conv3.sh $O/dgrad_fused > $O/dgrad_fused.log 2>&1
cp $O/dgrad_fused/pmc_summary.json $O/r04_pmc_conv3_dgrad_fused_32ch_96cube.json
KERNEL=conv3_wgrad3 FILTER=conv3_wgrad_kernel KARGS="--cin 32 --cout 16 --size 96 --batch 2" FAMILY="3x3x3 conv weight-grad" \
  SHAPE="dw[16,32,27] + dw3[16,32] @ 96^3 x 2" bash $ROOT/tools/pmc_conv3.sh $O/wgrad3 > $O/wgrad3.log 2>&1
cp $O/wgrad3/pmc_summary.json $O/r04_pmc_conv3_wgrad3_32to16_96cube.json
KERNEL=conv3_fused FILTER=conv3_fwd_pipe_kernel KARGS="--cin 16 --cout 16 --size 96 --batch 2" FAMILY="3x3x3 conv fwd + data-grad" \
  SHAPE="conv2 16->16 + IN sums @ 96^3 x 2" bash $ROOT/tools/pmc_conv3.sh $O/fwd16 > $O/fwd16.log 2>&1
cp $O/fwd16/pmc_summary.json $O/r04_pmc_conv3_fwd_16to16_96cube.json
echo "conv pmc done"
