#!/bin/bash
# per-kernel stats of the 160^3 step with encoder checkpointing (bench.py --config c4): bash tools/prof_c4.sh -> gpurun_out/c4_kernel_stats.csv
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_c4
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --config c4 --steps 5 --warmup 2 --windows 1 --no-cpu-baseline --no-roofline > $O/stats.log 2>&1
cp $O/*/*kernel_stats.csv gpurun_out/c4_kernel_stats.csv
echo c4 done
