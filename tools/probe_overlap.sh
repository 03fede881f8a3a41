# AdamW on a side-stream branch of the captured step (bench.py --overlap-update 1) for several caps on the optimizer kernel's grid
for g in 4096 1024 512 256 128; do
  echo "== UNETR_ADAMW_GRID=$g"
  UNETR_ADAMW_GRID=$g python bench.py --no-cpu-baseline --no-roofline --overlap-update 1 --windows 3 2>&1 | grep "timed region"
done
echo "== plain single-stream step"; python bench.py --no-cpu-baseline --no-roofline --overlap-update 0 --windows 3 2>&1 | grep "timed region"
for g in 256 128; do echo "== plain, UNETR_ADAMW_GRID=$g"; UNETR_ADAMW_GRID=$g python bench.py --no-cpu-baseline --no-roofline --overlap-update 0 --windows 3 2>&1 | grep "timed region"; done
