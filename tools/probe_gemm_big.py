"""Per-launch time and TFLOP/s of the bf16 GEMM forms at many-row encoder shapes (M = PROBE_M, default 6912 = batch 32 at 96^3):
the 256 x 256 ping-pong kernel (UNETR_GEMM_CFG=256) against the 128 x 128 tile (128), each as 10 back-to-back launches replayed
from a hipGraph.  Output tensors as the encoder forward writes them (bf16 only for qkv / linear1, fp32 + residual for the others)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
Fn = pkg.functional
dev = torch.device("cuda:0")
M, H, MLP = int(os.environ.get("PROBE_M", 6912)), 768, 3072


def timeit(fn, reps=10, iters=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / reps * 1e3


def main():
    torch.manual_seed(0)
    xb = torch.randn(M, H, device=dev).bfloat16()
    hb = torch.randn(M, MLP, device=dev).bfloat16()
    w = {n: (torch.randn(s, device=dev) * 0.02).bfloat16() for n, s in (("qkv", (3 * H, H)), ("p", (H, H)), ("w1", (MLP, H)), ("w2", (H, MLP)), ("pe", (H, 4096)))}
    pb = torch.randn(M, 4096, device=dev).bfloat16()
    x = torch.randn(M, H, device=dev)
    bias1, bias3 = torch.zeros(H, device=dev), torch.zeros(MLP, device=dev)
    qkvb = torch.empty(M, 3 * H, device=dev, dtype=torch.bfloat16)
    ab = torch.empty(M, MLP, device=dev, dtype=torch.bfloat16)
    x1 = torch.empty(M, H, device=dev)
    cases = [
        ("qkv   N=2304 K=768  (bf16 out)", 2 * M * 3 * H * H, lambda: Fn.gemm_bf16(xb, w["qkv"], M, 3 * H, H, Cb=qkvb)),
        ("mlp1  N=3072 K=768  (+bias, gelu, bf16 out)", 2 * M * MLP * H, lambda: Fn.gemm_bf16(xb, w["w1"], M, MLP, H, Cb=ab, bias=bias3, act=1)),
        ("proj  N=768  K=768  (+bias, res, fp32 out)", 2 * M * H * H, lambda: Fn.gemm_bf16(xb, w["p"], M, H, H, C=x1, bias=bias1, res=x, ldr=H)),
        ("mlp2  N=768  K=3072 (+bias, res, fp32 out)", 2 * M * H * MLP, lambda: Fn.gemm_bf16(hb, w["w2"], M, H, MLP, C=x1, bias=bias1, res=x, ldr=H)),
        ("patch N=768  K=4096 (+bias, fp32 out)", 2 * M * H * 4096, lambda: Fn.gemm_bf16(pb, w["pe"], M, H, 4096, C=x1, bias=bias1)),
    ]
    for cfg in os.environ.get("PROBE_CFGS", "0,128,256:4,256:3,256:2").split(","):
        os.environ.pop("UNETR_GEMM_CFG", None)
        os.environ.pop("UNETR_GEMM_BIG_WN", None)
        if cfg != "0":
            os.environ["UNETR_GEMM_CFG"] = cfg.split(":")[0]
            if ":" in cfg:
                os.environ["UNETR_GEMM_BIG_WN"] = cfg.split(":")[1]
        for name, flops, fn in cases:
            us = timeit(fn)
            print(f"cfg {cfg:>5}  {us:8.2f} us  {flops / us / 1e6:8.1f} TFLOP/s  {name}", flush=True)


if __name__ == "__main__":
    main()
