"""Per-launch cost of the ViT encoder's kernels at batch-2 shapes, each timed as 20 back-to-back launches replayed from a
hipGraph (kernel + launch boundary, the quantity the step actually pays).  Variants are selected through the tuning
environment hooks of the C ABI (read at launch time)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
Fn = pkg.functional
dev = torch.device("cuda:0")
M, H, MLP = int(os.environ.get("PROBE_M", 432)), 768, 3072


def timeit(fn, reps=20, iters=10):
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
    return e0.elapsed_time(e1) / iters / reps * 1e3      # us per launch


def env(**kw):
    for k in ("UNETR_GEMM_CFG", "UNETR_GEMM_SPLITS", "UNETR_GEMM_STAGES", "UNETR_LNGEMM_BN"):
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ[k] = str(v)


torch.manual_seed(0)
x = torch.randn(M, H, device=dev)
gam, bet = torch.ones(H, device=dev), torch.zeros(H, device=dev)
w = {n: (torch.randn(s, device=dev) * 0.02).bfloat16() for n, s in (("qkv", (3 * H, H)), ("p", (H, H)), ("w1", (MLP, H)), ("w2", (H, MLP)))}
bias3, bias1 = torch.zeros(MLP, device=dev), torch.zeros(H, device=dev)
xb = x.bfloat16()
hb = torch.randn(M, MLP, device=dev).bfloat16()
qkv = torch.empty(M, 3 * H, device=dev)
x1 = torch.empty(M, H, device=dev)
ab = torch.empty(M, MLP, device=dev, dtype=torch.bfloat16)
u = torch.empty(M, MLP, device=dev)
y1b = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
m1, r1 = torch.empty(M, device=dev), torch.empty(M, device=dev)
res = {}

env()
res["layernorm_fwd (bf16 out)"] = timeit(lambda: Fn.layernorm_fwd(x, gam, bet, bf16_out=y1b, want_fp32=False))
for cfg in ("6464", "64128", "6496"):
    env(UNETR_GEMM_CFG=cfg)
    res[f"gemm_bf16 qkv   N=2304 K=768  cfg {cfg}"] = timeit(lambda: Fn.gemm_bf16(xb, w["qkv"], M, 3 * H, H, C=qkv))
    res[f"gemm_bf16 mlp1  N=3072 K=768  cfg {cfg} (+gelu, bf16 out, pre)"] = timeit(
        lambda: Fn.gemm_bf16(xb, w["w1"], M, MLP, H, Cb=ab, bias=bias3, act=1, pre=u))
    qb2_ = torch.empty(M, 3 * H, device=dev, dtype=torch.bfloat16)
    res[f"gemm_bf16 qkv   N=2304 K=768  cfg {cfg} (bf16 out only)"] = timeit(lambda: Fn.gemm_bf16(xb, w["qkv"], M, 3 * H, H, Cb=qb2_))
for bn in (64, 128):
    env(UNETR_LNGEMM_BN=bn)
    res[f"ln_gemm   qkv   N=2304 K=768  BN {bn} (fp32 out, keeps xn/mean/rstd)"] = timeit(
        lambda: Fn.ln_gemm_bf16(x, gam, bet, w["qkv"], C=qkv, xn=y1b, mean=m1, rstd=r1))
    qb_ = torch.empty(M, 3 * H, device=dev, dtype=torch.bfloat16)
    res[f"ln_gemm   qkv   N=2304 K=768  BN {bn} (bf16 out only)"] = timeit(lambda: Fn.ln_gemm_bf16(x, gam, bet, w["qkv"], Cb=qb_))
    res[f"ln_gemm   mlp1  N=3072 K=768  BN {bn} (+gelu, bf16 out, pre, keeps)"] = timeit(
        lambda: Fn.ln_gemm_bf16(x, gam, bet, w["w1"], bias=bias3, act=1, Cb=ab, pre=u, xn=y1b, mean=m1, rstd=r1))
for cfg in ("6464", "6432", "3264"):
    for sp in (0, 1):
        env(UNETR_GEMM_CFG=cfg, **({"UNETR_GEMM_SPLITS": 1} if sp else {}))
        tag = "no split" if sp else "auto split"
        res[f"gemm_bf16 proj  N=768  K=768  cfg {cfg} {tag} (+bias,res)"] = timeit(
            lambda: Fn.gemm_bf16(xb, w["p"], M, H, H, C=x1, bias=bias1, res=x, ldr=H))
        res[f"gemm_bf16 mlp2  N=768  K=3072 cfg {cfg} {tag} (+bias,res)"] = timeit(
            lambda: Fn.gemm_bf16(hb, w["w2"], M, H, MLP, C=x1, bias=bias1, res=x, ldr=H))
for st in (2, 4, 6):
    env(UNETR_GEMM_CFG="6464", UNETR_GEMM_STAGES=st, UNETR_GEMM_SPLITS=1)
    res[f"gemm_bf16 mlp2  N=768  K=3072 cfg 6464 no split stages {st}"] = timeit(
        lambda: Fn.gemm_bf16(hb, w["w2"], M, H, MLP, C=x1, bias=bias1, res=x, ldr=H))
# data gradients (B read as [K,N])
dyb = torch.randn(M, H, device=dev).bfloat16()
dub = torch.randn(M, MLP, device=dev).bfloat16()
du = torch.empty(M, MLP, device=dev)
dub2 = torch.empty(M, MLP, device=dev, dtype=torch.bfloat16)
dx = torch.empty(M, H, device=dev)
for cfg in ("6464", "3264", "64128"):
    for sp in (0, 1):
        env(UNETR_GEMM_CFG=cfg, **({"UNETR_GEMM_SPLITS": 1} if sp else {}))
        tag = "no split" if sp else "auto split"
        res[f"dgrad du   N=3072 K=768  cfg {cfg} {tag} (gelu', fp32+bf16 out)"] = timeit(
            lambda: Fn.gemm_bf16(dyb, w["w2"], M, MLP, H, b_kn=True, C=du, Cb=dub2, act=2, aux=u, ldaux=MLP))
        res[f"dgrad dy2  N=768  K=3072 cfg {cfg} {tag}"] = timeit(lambda: Fn.gemm_bf16(dub, w["w1"], M, H, MLP, b_kn=True, C=dx))
        res[f"dgrad datt N=768  K=768  cfg {cfg} {tag}"] = timeit(lambda: Fn.gemm_bf16(dyb, w["p"], M, H, H, b_kn=True, C=dx))
env()
B, L, heads = M // 216 if M % 216 == 0 else 1, 216 if M % 216 == 0 else M, 12
att = None
res["attention_fwd (fp32 qkv)"] = timeit(lambda: Fn.attention_fwd(qkv, B, L, heads, 64, 1, out_bf16=y1b))
att, lse = Fn.attention_fwd(qkv, B, L, heads, 64, 1)
dout = torch.randn(M, H, device=dev)
dqb = torch.empty(M, 3 * H, device=dev, dtype=torch.bfloat16)
res["attention_bwd (2 kernels)"] = timeit(lambda: Fn.attention_bwd(qkv, att, dout, lse, B, L, heads, 64, 1, dqkv_bf16=dqb))
qkvb = qkv.bfloat16()
res["attention_bf16_fwd (bf16 qkv -> bf16 out)"] = timeit(lambda: Fn.attention_bf16_fwd(qkvb, B, L, heads, 64, y1b))
lse2 = Fn.attention_bf16_fwd(qkvb, B, L, heads, 64, y1b)
doutb = dout.bfloat16()
res["attention_bf16_bwd (2 kernels, bf16 only)"] = timeit(lambda: Fn.attention_bf16_bwd(qkvb, y1b, doutb, lse2, B, L, heads, 64))
dxb = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
res["layernorm_bwd (+dres, partials)"] = timeit(lambda: Fn.layernorm_bwd(dout, x, gam, m1, r1, dres=x1, dx_bf16=dxb))
for k, v in res.items():
    print(f"{v:8.2f} us  {k}")
