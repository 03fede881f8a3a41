# A/B of one environment switch on ONE box: bash tools/ab.sh VAR A_VALUE B_VALUE  (three interleaved rounds each)
V=$1; A=$2; B=$3
for r in 1 2 3; do
  for x in $A $B; do
    echo -n "$V=$x: "; env $V=$x python bench.py --no-cpu-baseline --no-roofline --windows 3 2>&1 | grep "timed region" | sed 's/.*done: //'
  done
done
