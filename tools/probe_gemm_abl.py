"""Timing ablations of the 256 x 256 GEMM kernel (UNETR_GEMM_BIG_ABL: 1 no DMA in the loop, 2 no fragment reads, 3 no MFMAs, 4 no
epilogue stores; results are wrong by construction, only the time matters)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
Fn = pkg.functional
dev = torch.device("cuda:0")
from tools.probe_gemm_big import timeit  # noqa: E402

os.environ["UNETR_GEMM_CFG"] = "256"
M = int(os.environ.get("PROBE_M", 6912))
for N, K in ((2304, 768), (2304, 3072), (768, 3072)):
    xb = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    yb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for ph in os.environ.get("PROBE_PH", "2,4").split(","):
        os.environ["UNETR_GEMM_BIG_PH"] = ph
        for abl in (0, 1, 2, 3, 4):
            os.environ["UNETR_GEMM_BIG_ABL"] = str(abl)
            us = timeit(lambda: Fn.gemm_bf16(xb, w, M, N, K, Cb=yb))
            print(f"N={N} K={K} phases {ph} abl {abl}: {us:8.2f} us  ({us / (K / 64):6.3f} us per K tile)", flush=True)
