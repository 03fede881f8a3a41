#!/bin/bash
# per-kernel stats of the batch-32 encoder forward under environment settings: bash tools/prof_encoder_big.sh <tag> [VAR=val ...]
# writes gpurun_out/<tag>_encb32_kernel_stats.csv (12 executions per kernel) and gpurun_out/<tag>_encb32.log (un-profiled time)
set -e
TAG=$1; shift
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for kv in "$@"; do export "$kv"; done
O=gpurun_out/prof_encb32_$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/prof_encoder_big.py > $O/run.log 2>&1
cp $O/*/*kernel_stats.csv gpurun_out/${TAG}_encb32_kernel_stats.csv
python3 tools/prof_encoder_big.py > gpurun_out/${TAG}_encb32.log 2>&1
tail -1 gpurun_out/${TAG}_encb32.log
