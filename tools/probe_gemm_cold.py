"""ViT GEMM shapes at batch 2 with COLD weights: every launch of the timed chain reads a different copy of the weight matrix
(enough copies to exceed L2 + Infinity Cache), as the real step does -- tools/probe_encoder.py re-reads one warm matrix.
Rows: ring depth (UNETR_GEMM_STAGES) x tile (UNETR_GEMM_CFG); us per launch inside a hipGraph chain."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
Fn = pkg.functional
dev = torch.device("cuda:0")
M, H, MLP = int(os.environ.get("PROBE_M", 432)), 768, 3072
NCOPY = int(os.environ.get("PROBE_COPIES", 96))


def timeit(fns, iters=5):
    for f in fns[:2]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / len(fns) * 1e3


def env(**kw):
    for k in ("UNETR_GEMM_CFG", "UNETR_GEMM_SPLITS", "UNETR_GEMM_STAGES"):
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ[k] = str(v)


torch.manual_seed(0)
xb = torch.randn(M, H, device=dev).bfloat16()
hb = torch.randn(M, MLP, device=dev).bfloat16()
x = torch.randn(M, H, device=dev)
o768, o2304, o3072 = (torch.empty(M, n, device=dev) for n in (H, 3 * H, MLP))
ob3072 = torch.empty(M, MLP, device=dev, dtype=torch.bfloat16)
bias1, bias3 = torch.zeros(H, device=dev), torch.zeros(MLP, device=dev)
W = {n: [(torch.randn(s, device=dev) * 0.02).bfloat16() for _ in range(NCOPY)]
     for n, s in (("qkv", (3 * H, H)), ("p", (H, H)), ("w1", (MLP, H)), ("w2", (H, MLP)))}
shapes = {
    "fwd qkv  N=2304 K=768 ": lambda w: Fn.gemm_bf16(xb, w, M, 3 * H, H, C=o2304),
    "fwd proj N=768  K=768 ": lambda w: Fn.gemm_bf16(xb, w, M, H, H, C=o768, bias=bias1, res=x, ldr=H),
    "fwd mlp1 N=3072 K=768 ": lambda w: Fn.gemm_bf16(xb, w, M, MLP, H, Cb=ob3072, bias=bias3, act=1, pre=o3072),
    "fwd mlp2 N=768  K=3072": lambda w: Fn.gemm_bf16(hb, w, M, H, MLP, C=o768, bias=bias1, res=x, ldr=H),
    "dgrad du   N=3072 K=768 ": lambda w: Fn.gemm_bf16(xb, w, M, MLP, H, b_kn=True, C=o3072, Cb=ob3072, act=2, aux=o3072, ldaux=MLP),
    "dgrad dy2  N=768  K=3072": lambda w: Fn.gemm_bf16(hb, w, M, H, MLP, b_kn=True, C=o768),
    "dgrad datt N=768  K=768 ": lambda w: Fn.gemm_bf16(xb, w, M, H, H, b_kn=True, C=o768),
    "dgrad dqkv N=768  K=2304": lambda w: Fn.gemm_bf16(hb[:, :3 * H].contiguous(), w, M, H, 3 * H, b_kn=True, C=o768),
}
wkey = {"fwd qkv": "qkv", "fwd proj": "p", "fwd mlp1": "w1", "fwd mlp2": "w2", "dgrad du": "w2", "dgrad dy2": "w1", "dgrad datt": "p", "dgrad dqkv": "qkv"}
hq = hb[:, :3 * H].contiguous()
shapes["dgrad dqkv N=768  K=2304"] = lambda w: Fn.gemm_bf16(hq, w, M, H, 3 * H, b_kn=True, C=o768)
cfgs = os.environ.get("PROBE_CFGS", "0,6464,6432,3264").split(",")
stages = os.environ.get("PROBE_STAGES", "0,6,8").split(",")
for name, fn in shapes.items():
    ws = W[wkey[" ".join(name.split()[:2])]]
    for cfg in cfgs:
        for st in stages:
            kw = {}
            if cfg != "0":
                kw["UNETR_GEMM_CFG"] = cfg
            if st != "0":
                kw["UNETR_GEMM_STAGES"] = st
            env(**kw)
            try:
                cold = timeit([lambda w=w: fn(w) for w in ws])
                warm = timeit([lambda: fn(ws[0])] * 24)
                print(f"{name}  cfg {cfg:>5s} stages {st:>2s}: cold {cold:6.2f} us   warm {warm:6.2f} us", flush=True)
            except Exception as e:
                print(f"{name}  cfg {cfg:>5s} stages {st:>2s}: {type(e).__name__} {str(e)[:60]}", flush=True)
