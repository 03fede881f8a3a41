#!/bin/bash
# per-kernel stats of the captured step in a precision mode: bash tools/prof_prec.sh <tag> <precision> -> gpurun_out/<tag>_kernel_stats.csv
set -e
TAG=$1; PREC=$2
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
O=gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --precision $PREC --steps 10 --warmup 3 --windows 1 --no-cpu-baseline --no-roofline > $O/stats.log 2>&1
cp $O/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
echo "$TAG done"
