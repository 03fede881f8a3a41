"""Micro-benchmarks of single hot kernels through the C ABI (for rocprofv3 --pmc runs and A/B timing).

    python tools/kernel_bench.py conv3_fwd --cin 16 --cout 16 --size 96 --batch 2 --prec bf16 --iters 20
    python tools/kernel_bench.py conv3_wgrad ... | gemm --m 432 --n 768 --k 3072 | instnorm ...

Prints one JSON line with the average launch time (HIP events on the launch stream) and the algorithmic
bytes / flops of the launch, so the PMC FETCH_SIZE / WRITE_SIZE of the same command can be compared.
"""
import argparse
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel", choices=["conv3_fwd", "conv3_dgrad", "conv3_wgrad", "conv3_fused", "conv3_dgrad_fused", "conv3_wgrad3", "gemm", "gemm_x3w", "gemm_dgrad", "gemm_wgrad", "gemm_bf16", "gemm_bf16_dgrad", "instnorm", "encoder_fwd", "tconv_fwd", "tconv_dgrad", "tconv_wgrad"])
    ap.add_argument("--cin", type=int, default=16)
    ap.add_argument("--cout", type=int, default=16)
    ap.add_argument("--size", type=int, default=96)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--m", type=int, default=432)
    ap.add_argument("--n", type=int, default=768)
    ap.add_argument("--k", type=int, default=3072)
    ap.add_argument("--prec", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bf16-out", action="store_true", help="gemm_bf16: bf16 output only (as the QKV projection of the step)")
    ap.add_argument("--graph", action="store_true", help="time one hipGraph holding `iters` launches (hides host launch cost)")
    a = ap.parse_args()
    pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
    Fn = pkg.functional
    dev = torch.device("cuda:0")
    prec = {"fp32": 0, "bf16": 1, "bf16x3": 2}[a.prec]
    g = torch.Generator(device="cpu").manual_seed(0)
    S, B = a.size, a.batch
    dims = (B, S, S, S)
    v = B * S ** 3
    adt = Fn.act_dtype(prec)            # feature maps are stored bf16 in bf16 mode
    esz = 2.0 if adt == torch.bfloat16 else 4.0
    if a.kernel.startswith("conv3"):
        x = torch.randn(B, S, S, S, a.cin, generator=g).to(dev).to(adt)
        w = (torch.randn(a.cout, a.cin, 3, 3, 3, generator=g) * 0.1).to(dev)
        dy = torch.randn(B, S, S, S, a.cout, generator=g).to(dev).to(adt)
        flops = 2.0 * v * a.cin * a.cout * 27
        nbytes = esz * v * (a.cin + a.cout) + 4.0 * w.numel()
        if a.kernel == "conv3_fused":     # residual-block front: 3x3x3 conv + InstanceNorm sums + 1x1x1 conv on the same window
            w3 = (torch.randn(a.cout, a.cin, 1, 1, 1, generator=g) * 0.2).to(dev)
            fn = lambda: Fn.conv3_fused(x, a.cin, w, w3, (B, S, S, S), prec)
            nbytes = esz * v * (a.cin + 2 * a.cout) + 4.0 * (w.numel() + w3.numel())
            flops = 2.0 * v * a.cin * a.cout * 28
            label = f"conv3_fused(+IN sums +1x1) {a.cin}->{a.cout} @ {S}^3 B={B} {a.prec}"
        elif a.kernel == "conv3_dgrad_fused":  # residual-block input gradient: conv3x3x3^T(dc1) + conv1x1x1^T(dc3) in one launch
            w3 = (torch.randn(a.cout, a.cin, 1, 1, 1, generator=g) * 0.2).to(dev)
            dy3 = torch.randn(B, S, S, S, a.cout, generator=g).to(dev).to(adt)
            dxo = torch.empty(B, S, S, S, a.cin, device=dev, dtype=adt)
            fn = lambda: Fn.conv3_dgrad_fused(dy, dy3, w, w3, dxo, dims, prec)
            nbytes = esz * v * (a.cin + 2 * a.cout) + 4.0 * (w.numel() + w3.numel())
            flops = 2.0 * v * a.cin * a.cout * 28
        elif a.kernel == "conv3_wgrad3":       # weight gradients of the 3x3x3 conv and of the 1x1x1 branch sharing its input
            dy3 = torch.randn(B, S, S, S, a.cout, generator=g).to(dev).to(adt)
            dw3 = torch.empty(a.cout, a.cin, 1, 1, 1, device=dev)
            fn = lambda: Fn.conv3_wgrad(x, a.cin, dy, a.cout, dims, a.cin, a.cout, prec, dy3=dy3, out3=dw3)
            nbytes = esz * v * (a.cin + 2 * a.cout) + 4.0 * w.numel()
            flops = 2.0 * v * a.cin * a.cout * 28
        elif a.kernel == "conv3_fwd":
            fn = lambda: Fn.conv3(x, a.cin, w, dims, prec)
        elif a.kernel == "conv3_dgrad":
            fn = lambda: Fn.conv3(dy, a.cout, w, dims, prec, mode=1)
        else:
            fn = lambda: Fn.conv3_wgrad(x, a.cin, dy, a.cout, dims, a.cin, a.cout, prec)
        label = f"{a.kernel} {a.cin}->{a.cout} @ {S}^3 B={B} {a.prec}"
    elif a.kernel.startswith("gemm"):
        M, N, K = a.m, a.n, a.k
        x = torch.randn(M, K, generator=g).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
        dy = torch.randn(M, N, generator=g).to(dev)
        flops = 2.0 * M * N * K
        nbytes = 4.0 * (M * K + N * K + M * N)
        xb, wb, dyb = x.bfloat16(), w.bfloat16(), dy.bfloat16()
        yo, dxo = torch.empty(M, N, device=dev), torch.empty(M, K, device=dev)
        yob = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        if a.kernel.startswith("gemm_bf16"):
            nbytes = 2.0 * (M * K + N * K) + (2.0 if a.bf16_out else 4.0) * M * N
        ww = torch.empty(N, K, dtype=torch.int32, device=dev)          # bf16x3: the weight as pre-split words (the optimizer-maintained shadow)
        if a.kernel == "gemm_x3w":
            Fn._split_words(w, ww, N * K)
        fn = {"gemm_bf16": (lambda: Fn.gemm_bf16(xb, wb, M, N, K, Cb=yob)) if a.bf16_out else (lambda: Fn.gemm_bf16(xb, wb, M, N, K, C=yo)),
              "gemm_x3w": lambda: Fn.gemm(x, w, yo, M, N, K, lda=K, ldb=K, ldc=N, prec=2, b_words=ww),
              "gemm_bf16_dgrad": lambda: Fn.gemm_bf16(dyb, wb, M, K, N, b_kn=True, C=dxo),
              "gemm": lambda: Fn.linear_fwd(x, w, None, prec), "gemm_dgrad": lambda: Fn.linear_dgrad(dy, w, prec),
              "gemm_wgrad": lambda: Fn.linear_wgrad(dy, x, prec)}[a.kernel]
        label = f"{a.kernel} M={M} N={N} K={K} {a.prec}"
    elif a.kernel.startswith("tconv"):
        x = torch.randn(B, S, S, S, a.cin, generator=g).to(dev).to(adt)
        w = (torch.randn(a.cin, a.cout, 2, 2, 2, generator=g) * 0.1).to(dev)
        dy = torch.randn(B, 2 * S, 2 * S, 2 * S, a.cout, generator=g).to(dev).to(adt)
        dims = (B, S, S, S)
        flops = 2.0 * v * a.cin * a.cout * 8
        nbytes = esz * v * (a.cin + 8 * a.cout)
        fn = {"tconv_fwd": lambda: Fn.tconv_fwd(x, a.cin, w, dims, a.cin, a.cout, prec)[0],
              "tconv_dgrad": lambda: Fn.tconv_dgrad(dy, a.cout, w, dims, a.cin, a.cout, prec),
              "tconv_wgrad": lambda: Fn.tconv_wgrad(x, a.cin, dy, a.cout, dims, a.cin, a.cout, prec)}[a.kernel]
        label = f"{a.kernel} {a.cin}->{a.cout} @ {S}^3 B={B} {a.prec}"
    elif a.kernel == "instnorm":
        x = torch.randn(B, S, S, S, a.cout, generator=g).to(dev).to(adt)
        flops, nbytes = 0.0, esz * v * a.cout
        fn = lambda: Fn.instnorm_stats(x, a.cout, B, S ** 3, a.cout)
        label = f"instnorm_stats C={a.cout} @ {S}^3 B={B}"
    else:  # encoder_fwd: ViT encoder forward (patch-embed + 12 blocks + final norm) at batch B
        cfg = dict(in_channels=1, out_channels=4, img_size=(96, 96, 96), feature_size=16, hidden_size=768, mlp_dim=3072,
                   num_heads=12, pos_embed="perceptron", norm_name="instance", res_block=True)
        model = pkg.UNETR(**cfg).to(dev)
        model.precision = a.prec
        xin = torch.rand(B, 1, 96, 96, 96, generator=g).to(dev)
        pe = model.vit.patch_embedding

        def fn():
            with torch.no_grad():
                z = Fn.PatchEmbedFn.apply(xin, pe.patch_embeddings[1].weight, pe.patch_embeddings[1].bias, pe.position_embeddings, 16, prec)
                blocks = list(model.vit.blocks)
                for i, blk in enumerate(blocks):
                    nxt = blocks[i + 1].norm1 if i + 1 < len(blocks) else None
                    z = Fn.TransformerBlockFn.apply(
                        z, blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.out_proj.weight, blk.attn.out_proj.bias,
                        blk.norm2.weight, blk.norm2.bias, blk.mlp.linear1.weight, blk.mlp.linear1.bias, blk.mlp.linear2.weight,
                        blk.mlp.linear2.bias, B, 216, 12, prec, False,
                        None if nxt is None else nxt.weight.detach(), None if nxt is None else nxt.bias.detach())
                return Fn.LayerNormFn.apply(z, model.vit.norm.weight, model.vit.norm.bias)
        flops = 39.771e9 * B       # encoder forward per volume
        nbytes = 4.0 * 88341504    # fp32 encoder weights streamed once
        label = f"ViT encoder forward B={B} {a.prec}"
    for _ in range(a.warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if a.graph:
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(a.iters):
                fn()
        gr.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters / 5
    else:
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
    print(json.dumps({"kernel": label, "ms": round(ms, 4), "GB/s": round(nbytes / ms / 1e6, 1), "TFLOP/s": round(flops / ms / 1e9, 2),
                      "algorithmic_bytes": nbytes, "algorithmic_flops": flops}))


if __name__ == "__main__":
    main()
