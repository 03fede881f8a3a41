import torch


def relerr(a, b):
    """max |a-b| / max |b| on CPU float64"""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


TOL = {0: 2e-5, 1: 3e-2, 2: 1e-4}  # fp32 MFMA (exact fma chains, different summation order) / bf16 operands / bf16x3 (split operands, ~16-bit products)
