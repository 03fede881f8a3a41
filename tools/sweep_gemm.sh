# sweep tile config / split-K for the batch-2 encoder GEMM shapes (run on the GPU box)
for shape in "--m 432 --n 768 --k 3072" "--m 432 --n 3072 --k 768" "--m 432 --n 2304 --k 768" "--m 432 --n 768 --k 768"; do
  for cfg in 64 128; do
    for sp in 1 2 3 4 6 8 12; do
      for kern in gemm gemm_dgrad; do
        r=$(UNETR_GEMM_CFG=$cfg UNETR_GEMM_SPLITS=$sp python3 tools/kernel_bench.py $kern $shape --iters 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms']*1000)")
        echo "$kern $shape cfg=$cfg splits=$sp us=$r"
      done
    done
  done
done
