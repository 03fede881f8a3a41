#!/bin/bash
# per-kernel stats of the captured step under environment settings: bash tools/prof_env.sh <tag> [VAR=val ...]
# writes gpurun_out/<tag>_kernel_stats.csv (27 executions per kernel: divide by 27)
set -e
TAG=$1; shift
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for kv in "$@"; do export "$kv"; done
O=gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 20 --warmup 5 --windows 1 --no-cpu-baseline --no-roofline > $O/stats.log 2>&1
cp $O/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
echo "$TAG done"
