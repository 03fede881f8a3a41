# A/B of one bench.py command-line switch on ONE box: bash tools/ab_flag.sh --flag A_VALUE B_VALUE  (three interleaved rounds each)
F=$1; A=$2; B=$3
for r in 1 2 3; do
  for x in $A $B; do
    echo -n "$F $x: "; python bench.py $F $x --no-cpu-baseline --no-roofline --windows 3 2>&1 | grep "timed region" | sed 's/.*done: //'
  done
done
