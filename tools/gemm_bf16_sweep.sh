#!/bin/bash
# bf16-storage GEMM on the encoder shapes (graph-timed): ring depth sweep via UNETR_GEMM_STAGES
for shape in "432 2304 768" "432 768 768" "432 3072 768" "432 768 3072"; do
  set -- $shape
  for ns in 2 4 6; do
    for k in gemm_bf16 gemm_bf16_dgrad; do
      echo -n "NS=$ns "; UNETR_GEMM_STAGES=$ns timeout -k 10 120 python tools/kernel_bench.py $k --m $1 --n $2 --k $3 --iters 20 --graph 2>/dev/null | cut -c1-100
    done
  done
done
for shape in "13824 2304 768" "13824 3072 768" "13824 768 3072"; do
  set -- $shape
  for ns in 2 3 4; do
    for k in gemm_bf16 gemm_bf16_dgrad; do
      echo -n "NS=$ns "; UNETR_GEMM_STAGES=$ns timeout -k 10 120 python tools/kernel_bench.py $k --m $1 --n $2 --k $3 --iters 20 --graph 2>/dev/null | cut -c1-100
    done
  done
done
