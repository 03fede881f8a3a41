#!/bin/bash
# Round profile set (run on the GPU box from the repo root): per-kernel stats of the captured step, HBM traffic per kernel (two
# --pmc passes), SQ counters for the dominant GEMM kernels.  Outputs under gpurun_out/; copy the summaries into profiles/.
#   usage: bash tools/collect_profiles.sh r03
set -e
TAG=${1:-r03}
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out/${TAG}_profiles
rm -rf $O && mkdir -p $O
# 1. kernel stats of the whole step replayed as one hipGraph (27 executions: 2 eager warm-up + 5 warm-up + 20 timed)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 5 --windows 1 --no-cpu-baseline --no-roofline > $O/stats.log 2>&1
cp $O/stats/*/*kernel_stats.csv $O/${TAG}_hipgraph_kernel_stats.csv
echo "stats done"
# 2. HBM traffic per kernel of the step
PMC_STEP_NAME=${TAG}_pmc_step_kernels.json bash tools/pmc_step.sh > $O/pmc_step.log 2>&1
cp gpurun_out/${TAG}_pmc_step_kernels.json $O/
echo "pmc step done"
# 3. SQ counters (MFMA busy, waits, instruction mix) + HBM bytes: the batch-2 QKV GEMM (dominant kernel of the step) and the
#    large-tile kernel at 6912 rows
KERNEL=gemm_bf16 FILTER=gemm_bf16_kernel KARGS="--m 432 --n 2304 --k 768 --bf16-out" FAMILY="ViT Linear GEMMs" SHAPE="qkv 432x2304x768" bash tools/pmc_conv3.sh $ROOT/$O/pmc_gemm_qkv_b2 > $O/pmc_gemm_qkv_b2.log 2>&1
cp $O/pmc_gemm_qkv_b2/pmc_summary.json $O/${TAG}_pmc_gemm_bf16_qkv_432rows.json
KERNEL=gemm_bf16 FILTER=gemm_bf16_big_kernel KARGS="--m 6912 --n 2304 --k 768 --bf16-out" FAMILY="ViT Linear GEMMs (large tile)" SHAPE="qkv 6912x2304x768" bash tools/pmc_conv3.sh $ROOT/$O/pmc_gemm_qkv_b32 > $O/pmc_gemm_qkv_b32.log 2>&1
cp $O/pmc_gemm_qkv_b32/pmc_summary.json $O/${TAG}_pmc_gemm_bf16_big_qkv_6912rows.json
echo "pmc gemm done"
# 4. probe tables
python3 tools/probe_gemm_big.py > $O/${TAG}_probe_gemm_6912rows.txt 2>&1
python3 tools/probe_encoder.py > $O/${TAG}_probe_encoder_432rows.txt 2>&1
python3 tools/probe_gw.py > $O/${TAG}_probe_grouped_wgrad.txt 2>&1
echo "probes done"
