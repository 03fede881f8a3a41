#!/bin/bash
# HBM traffic per kernel of the benched step (rocprofv3 --pmc, one counter per pass as MI355X_MICROARCH.md prescribes: FETCH_SIZE
# and WRITE_SIZE cannot share a pass; --kernel-trace only, no other trace domain), summarised into profiles/r02_pmc_step_kernels.json:
#   hbm read = FETCH_SIZE x 1024 x 2 (gfx950 reports half the bytes of wide coalesced streams), write = WRITE_SIZE x 1024.
# Usage (on the GPU box, from the repo root):  bash tools/pmc_step.sh
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
[ -f bench.py ] || { echo "pmc_step.sh: bench.py not found in $ROOT" >&2; exit 1; }
OUT=gpurun_out/pmc_step
rm -rf $OUT && mkdir -p $OUT
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
python3 tools/pmc_step_summary.py $OUT "$CMD"
