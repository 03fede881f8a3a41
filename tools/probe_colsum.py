"""The grouped column-sum launch of the step in isolation (bias gradients of the 36 biased Linear layers from their data gradients,
LayerNorm gamma / beta from the per-block partial rows, position-embedding gradient), replayed from a hipGraph."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
capi = pkg._capi
from tools.probe_gemm_big import timeit  # noqa: E402

dev = torch.device("cuda:0")
M, H, MLP = 432, 768, 3072
sets = {"all": [], "bias": [], "ln": [], "pos": []}
keep = []


def prob(x, N, ld, rows):
    out = torch.empty(N, device=dev)
    keep.extend([x, out])
    return (x, out, ld, rows, N)


for _ in range(12):
    for p in (prob(torch.randn(M, H, device=dev), H, H, M), prob(torch.randn(M, H, device=dev), H, H, M),
              prob(torch.randn(M, MLP, device=dev).bfloat16(), MLP, MLP, M)):
        sets["bias"].append(p)
    for _ in range(2):
        part = torch.randn(108, 2 * H, device=dev)
        sets["ln"] += [prob(part, H, 2 * H, 108), prob(part[:, H:], H, 2 * H, 108)]
sets["pos"].append(prob(torch.randn(2, 216 * H, device=dev), 216 * H, 216 * H, 2))
sets["all"] = sets["bias"] + sets["ln"] + sets["pos"]
for name, ps in sets.items():
    arr = (capi.ColsumProblem * len(ps))()
    nbytes = 0
    for i, (x, out, ld, rows, N) in enumerate(ps):
        arr[i].x, arr[i].out, arr[i].ld, arr[i].M, arr[i].N, arr[i].x_bf16 = x.data_ptr(), out.data_ptr(), ld, rows, N, int(x.dtype == torch.bfloat16)
        nbytes += rows * N * x.element_size()
    us = timeit(lambda: capi.call("unetr_colsum_grouped", arr, len(ps), torch.cuda.current_stream().cuda_stream), reps=5)
    print(f"{name:5s}: {len(ps):3d} problems, {nbytes / 1e6:6.1f} MB, {us:7.1f} us  ({nbytes / us / 1e3:6.0f} GB/s)")
