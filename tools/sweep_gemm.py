"""Sweep tile config / split-K of the batch-2 encoder GEMM shapes; kernel time via hipGraph replay (no host cost)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
Fn = pkg.functional
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
shapes = [(432, 768, 3072), (432, 3072, 768), (432, 2304, 768), (432, 768, 768)]
REP = 20
for (M, N, K) in shapes:
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    dy = torch.randn(M, N, generator=g).to(dev)
    for kern, fn in (("fwd", lambda: Fn.linear_fwd(x, w, None, 1)), ("dgrad", lambda: Fn.linear_dgrad(dy, w, 1))):
        res = []
        for cfg in (64, 128):
            for sp in (1, 2, 3, 4, 6, 8, 12, 16):
                os.environ["UNETR_GEMM_CFG"] = str(cfg)
                os.environ["UNETR_GEMM_SPLITS"] = str(sp)
                fn()
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    for _ in range(REP):
                        fn()
                gr.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gr.replay()
                gr.replay()
                e1.record()
                torch.cuda.synchronize()
                res.append((e0.elapsed_time(e1) / (2 * REP) * 1000, cfg, sp))
        res.sort()
        print(f"{kern} M={M} N={N} K={K}: " + " | ".join(f"{t:.1f}us cfg{c} s{s}" for t, c, s in res[:5]) + f"  worst {res[-1][0]:.1f}", flush=True)
