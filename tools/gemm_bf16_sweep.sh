#!/bin/bash
# bf16-storage GEMM vs the fp32-storage GEMM on the encoder shapes (graph-timed)
for shape in "432 2304 768" "432 768 768" "432 3072 768" "432 768 3072" "13824 2304 768" "13824 768 768" "13824 3072 768" "13824 768 3072"; do
  set -- $shape
  for k in gemm gemm_bf16 gemm_dgrad gemm_bf16_dgrad; do
    # dgrad: reduction over N -> pass (m, n, k) so that the forward weight is [n, k]
    timeout -k 10 120 python tools/kernel_bench.py $k --m $1 --n $2 --k $3 --iters 20 --graph | cut -c1-120
  done
done
