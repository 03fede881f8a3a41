# per-family kernel time of the step under an environment switch: bash tools/ab_family.sh VAR A B "family substring"
V=$1; A=$2; B=$3; F=$4
for r in 1 2; do
  for x in $A $B; do
    env $V=$x python bench.py --no-cpu-baseline --windows 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V=$x', d['ms_per_step'], [(f['family'][:28], round(f['ms_per_step'],4)) for f in d['families'] if '$F' in f['family']])"
  done
done
