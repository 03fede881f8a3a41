# A/B of one environment switch on ONE box, three interleaved rounds: bash tools/ab_env.sh VAR A B [C ...]
# prints ms/step (median of 3 windows) and the per-family kernel times whose name contains $FAM (default: all of them)
V=$1; shift
for r in 1 2 3; do
  for x in "$@"; do
    env $V=$x python bench.py --no-cpu-baseline --windows 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
fam='${FAM:-}'
print('$V=$x', d['ms_per_step'], 'launches', sum(f.get('launches_per_step',0) for f in d['families']), [(f['family'][:22], round(f['ms_per_step'],4), f.get('launches_per_step')) for f in d['families'] if fam in f['family']][:6])"
  done
done
