"""The grouped ViT weight-gradient launch alone (48 Linear problems of the 12 blocks at M token rows, operands warm in the MALL)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
capi = pkg._capi
from tools.probe_gemm_big import timeit  # noqa: E402

dev = torch.device("cuda:0")
M, H, MLP = int(os.environ.get("PROBE_M", 432)), 768, 3072
shapes = [(3 * H, H), (H, H), (MLP, H), (H, MLP)] * 12
arr = (capi.GroupedProblem * len(shapes))()
keep = []
flops = 0
for i, (N, K) in enumerate(shapes):
    dy = torch.randn(M, N, device=dev).bfloat16()
    x = torch.randn(M, K, device=dev).bfloat16()
    out = torch.empty(N, K, device=dev)
    keep += [dy, x, out]
    arr[i].dy, arr[i].x, arr[i].dw, arr[i].M, arr[i].N, arr[i].K = dy.data_ptr(), x.data_ptr(), out.data_ptr(), M, N, K
    flops += 2 * M * N * K
us = timeit(lambda: capi.call("unetr_gemm_bf16_grouped_wgrad", arr, len(shapes), torch.cuda.current_stream().cuda_stream), reps=3)
print(f"plain gradient store (32-token stages x 2, row-coalesced epilogue): {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  "
      f"({sum(n * k for n, k in shapes) * 4 / us / 1e3:6.1f} GB/s of dW stores)", flush=True)
