"""GPU parity of every HIP kernel against a plain PyTorch fp32 CPU reference of the same op.
Everything goes through the C ABI (ctypes) exactly as the product path does."""
import math

import pytest
import torch
import torch.nn.functional as F

from util import TOL, relerr

pytestmark = pytest.mark.gpu


def g(*shape, seed=0, scale=1.0):
    gen = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=gen) * scale


def rq(t, prec):
    """what the kernels see of a feature map in precision mode `prec`: bf16 mode stores feature maps as bf16"""
    return t.bfloat16().float() if prec == 1 else t


def act(t, prec, dev):
    """feature map on the device in the activation storage type of `prec` (include/unetr_hip.h, ACTIVATION STORAGE)"""
    return t.to(dev).bfloat16() if prec == 1 else t.to(dev)


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(432, 768, 768), (37, 50, 44), (432, 2304, 768), (8, 128, 4096), (1000, 16, 32), (300, 32, 256),
                                   (4096, 128, 64), (16, 300, 5000), (16, 16, 70000), (16, 40, 3000), (24, 200, 2000)])
def test_gemm_nt_nn_tn(pkg, dev, prec, M, N, K):
    Fn = pkg.functional
    x, w = g(M, K, seed=1), g(N, K, seed=2)
    ref = x @ w.t()
    y = Fn.linear_fwd(x.to(dev), w.to(dev), None, prec)
    assert relerr(y, ref) < TOL[prec]
    dy = g(M, N, seed=3)
    assert relerr(Fn.linear_dgrad(dy.to(dev), w.to(dev), prec), dy @ w) < TOL[prec]
    assert relerr(Fn.linear_wgrad(dy.to(dev), x.to(dev), prec), dy.t() @ x) < TOL[prec]


@pytest.mark.parametrize("M,N,K", [(432, 768, 768), (432, 2304, 768), (432, 768, 3072), (432, 3072, 768), (430, 772, 96), (33, 64, 32), (1500, 3072, 768),
                                   (1000, 768, 3072), (864, 2304, 768), (216, 128, 64), (50, 200, 160)])
def test_gemm_bf16x3_dma(pkg, dev, monkeypatch, M, N, K):
    """bf16x3 Linear GEMMs on the LDS-DMA kernel (csrc/gemm_bf16.hip, X3 instantiations: raw fp32 through LDS, operands split in
    registers at fragment time): forward with every epilogue and the [K,N]-operand data gradient (four ds_read_b32 per chunk), split-K
    shapes, ragged M / N, 1 to 96 stages, the 128 x 128 tile at many rows -- against fp64 products and against the generic
    fp32-storage family it replaces (UNETR_X3_GEMM_DMA=0)."""
    Fn = pkg.functional
    x, w, dy = g(M, K, seed=1), g(N, K, seed=2, scale=0.1), g(M, N, seed=3)
    b, res, u = g(N, seed=4), g(M, N, seed=5), g(M, K, seed=6)
    lin = (x.double() @ w.double().t())
    ur = u.clone().requires_grad_(True)
    F.gelu(ur).sum().backward()
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("UNETR_X3_GEMM_DMA", mode)
        xd, wd, dyd = x.to(dev), w.to(dev), dy.to(dev)
        y0 = Fn.linear_fwd(xd, wd, None, 2)
        pre = torch.empty(M, N, device=dev)
        y1 = Fn.linear_fwd(xd, wd, b.to(dev), 2, res=res.to(dev), act=1, pre=pre)
        dx0 = Fn.linear_dgrad(dyd, wd, 2)
        dx1 = Fn.linear_dgrad(dyd, wd, 2, aux=u.to(dev))
        acc = torch.ones(M, N, device=dev)
        Fn.gemm(xd, wd, acc, M, N, K, lda=K, ldb=K, ldc=N, prec=2, accumulate=True, alpha=0.5)
        outs[mode] = [t.cpu() for t in (y0, pre, y1, dx0, dx1, acc)]
    refs = [lin.float(), (lin + b).float(), (F.gelu(lin + b) + res).float(), (dy.double() @ w.double()).float(),
            ((dy.double() @ w.double()) * ur.grad.double()).float(), (1 + 0.5 * lin).float()]
    for k, (a, o, r) in enumerate(zip(outs["1"], outs["0"], refs)):
        assert relerr(a, r) < TOL[2], k
        assert relerr(a, o) < TOL[2], k


@pytest.mark.parametrize("M,N,K", [(432, 768, 768), (432, 2304, 768), (432, 768, 3072), (37, 56, 64), (2000, 384, 128),
                                   (1500, 3072, 768), (8, 128, 4096), (130, 200, 192)])
def test_gemm_bf16_storage(pkg, dev, M, N, K):
    """LDS-DMA GEMM on bf16-stored operands: both B layouts, fp32 and bf16 outputs, exact against the same bf16 inputs
    multiplied in fp64 (only the fp32 accumulation order differs)."""
    Fn = pkg.functional
    x, w, dy = g(M, K, seed=1).bfloat16(), g(N, K, seed=2).bfloat16(), g(M, N, seed=3).bfloat16()
    ref = (x.double() @ w.double().t()).float()
    xd, wd, dyd = x.to(dev), w.to(dev), dy.to(dev)
    y = torch.empty(M, N, device=dev)
    yb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    Fn.gemm_bf16(xd, wd, M, N, K, C=y, Cb=yb)
    assert relerr(y, ref) < 2e-5
    assert torch.equal(yb.cpu(), y.cpu().bfloat16())
    if N % 64 == 0 and K % 8 == 0:   # data gradient: dx[M,K] = dy[M,N] @ w[N,K], reduction over N, B read as [N,K] = [K_red, N_out]
        dx = torch.empty(M, K, device=dev)
        Fn.gemm_bf16(dyd, wd, M, K, N, b_kn=True, C=dx)
        assert relerr(dx, (dy.double() @ w.double()).float()) < 2e-5


@pytest.mark.parametrize("M,N,K,force", [(6912, 2304, 768, 0), (6912, 3072, 768, 0), (6912, 768, 3072, 0), (1030, 520, 192, 4), (1024, 512, 64, 4),
                                         (2100, 772, 128, 4), (1500, 1026, 320, 4), (300, 256, 3072, 4), (1030, 520, 192, 3), (2100, 772, 128, 3),
                                         (1500, 1026, 320, 2), (1024, 384, 64, 3), (1300, 130, 256, 2),
                                         (1024, 512, 64, 2), (2100, 772, 128, 2), (1030, 520, 192, 2), (300, 256, 3072, 2)])
def test_gemm_bf16_big_tile(pkg, dev, monkeypatch, M, N, K, force):
    """The 256 x {256, 192, 128} ping-pong kernel (8 waves, two groups one barrier interval apart, LDS-DMA into half-tile regions):
    picked by itself at encoder shapes of batch 32 (N = 2304 / 3072 / 768 -> tile widths 256 / 192 / 128), forced (UNETR_GEMM_CFG=256,
    UNETR_GEMM_BIG_WN) on ragged shapes -- M / N tails, 1 / 2 / 3 /
    5 / 48 K tiles (odd and even counts walk both LDS buffers), N % 4 != 0 (scalar epilogue) -- with every epilogue kind,
    against fp64 products of the same bf16 inputs."""
    Fn = pkg.functional
    if force:                       # force = columns / 64 of the tile (256 x 256 / 192 / 128); 0 = the dispatcher's own choice
        monkeypatch.setenv("UNETR_GEMM_CFG", "256")
        monkeypatch.setenv("UNETR_GEMM_BIG_WN", str(force))
    L = 206 if M == 1030 else M
    x, w = g(M, K, seed=1).bfloat16(), g(N, K, seed=2, scale=0.1).bfloat16()
    b, res, aux = g(N, seed=3), g(L, N, seed=4), g(M, N, seed=6)
    xd, wd = x.to(dev), w.to(dev)
    lin = (x.double() @ w.double().t()).float()
    y = torch.full((M, N), float("nan"), device=dev)
    yb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    Fn.gemm_bf16(xd, wd, M, N, K, C=y, Cb=yb)
    assert relerr(y, lin) < 2e-5
    assert torch.equal(yb.cpu(), y.cpu().bfloat16())
    Fn.gemm_bf16(xd, wd, M, N, K, C=y, bias=b.to(dev), res=res.to(dev), ldr=N, res_mod=L)
    assert relerr(y, lin + b + res.repeat(M // L, 1)) < 2e-5
    pre = torch.empty(M, N, device=dev)
    Fn.gemm_bf16(xd, wd, M, N, K, Cb=yb, bias=b.to(dev), act=1, pre=pre)
    assert relerr(pre, lin + b) < 2e-5 and relerr(yb.float(), F.gelu(lin + b)) < 5e-3
    auxr = aux.clone().requires_grad_(True)
    F.gelu(auxr).sum().backward()
    Fn.gemm_bf16(xd, wd, M, N, K, C=y, act=2, aux=aux.to(dev), ldaux=N)
    assert relerr(y, lin * auxr.grad) < 2e-5
    y0 = g(M, N, seed=7)
    y = y0.to(dev)
    Fn.gemm_bf16(xd, wd, M, N, K, C=y, accumulate=True, alpha=0.5)
    assert relerr(y, y0 + 0.5 * lin) < 2e-5


def test_gemm_bf16_grouped_wgrad(pkg, dev):
    """grouped dW_i = dY_i^T X_i on bf16-stored operands (both read through transposing LDS loads; 432 tokens = a ragged
    last 64-token stage; ragged N / K tiles) against fp64 products of the same bf16 inputs"""
    capi = pkg._capi
    shapes = [(432, 768, 768), (432, 2304, 768), (432, 768, 3072), (216, 200, 136), (64, 8, 8), (1000, 384, 128)]
    arr = (capi.GroupedProblem * len(shapes))()
    keep, refs, outs = [], [], []
    for i, (M, N, K) in enumerate(shapes):
        dy, x = g(M, N, seed=10 + i).bfloat16(), g(M, K, seed=20 + i).bfloat16()
        dyd, xd, out = dy.to(dev), x.to(dev), torch.full((N, K), float("nan"), device=dev)
        keep += [dyd, xd]
        outs.append(out)
        refs.append((dy.double().t() @ x.double()).float())
        arr[i].dy, arr[i].x, arr[i].dw, arr[i].M, arr[i].N, arr[i].K = dyd.data_ptr(), xd.data_ptr(), out.data_ptr(), M, N, K
    capi.call("unetr_gemm_bf16_grouped_wgrad", arr, len(shapes), torch.cuda.current_stream().cuda_stream)
    for out, ref, shp in zip(outs, refs, shapes):
        assert relerr(out, ref) < 2e-5, shp


def test_grouped_wgrad_fused_epilogues(pkg, dev):
    """The grouped weight-gradient launch with (a) AdamW in its epilogue (unetr_gemm_bf16_grouped_wgrad_adamw) + the table-driven
    AdamW over the other arena ranges (unetr_adamw_ranges), and (b) bf16 gradients written into a communication buffer
    (unetr_gemm_bf16_grouped_wgrad_bf16out) + the table-driven cast (unetr_cast_bf16_ranges): bit for bit what the plain launch
    followed by unetr_adamw / unetr_cast_bf16 produces (torch.nn.Linear backward + torch.optim.AdamW.step,
    unetr_segmentation_3d.py:224-225).  Ragged tiles (N, K not multiples of 128), a ragged token count, parameters between the
    weight matrices that only the range kernels touch, two optimizer steps with per-parameter step counts."""
    import ctypes
    capi = pkg._capi
    st = torch.cuda.current_stream().cuda_stream
    shapes = [(432, 768, 256), (216, 200, 136), (64, 8, 8), (432, 128, 384)]
    # arena: [bias-like 40][W0][48][W1][8][W2][W3][24]; every slice a multiple of 8 elements
    sizes, is_w = [40], [False]
    for k, (_, N, K) in enumerate(shapes):
        sizes += [N * K, (48, 8, 0, 24)[k]]
        is_w += [True, False]
    sizes, is_w = zip(*[(n, w) for n, w in zip(sizes, is_w) if n])
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += n
    nparam = len(sizes)
    torch.manual_seed(5)
    p0 = (torch.randn(total) * 0.05).to(dev)
    grads_other = (torch.randn(total) * 0.01).to(dev)
    ops = [(g(M, N, seed=30 + i).bfloat16().to(dev), g(M, K, seed=40 + i).bfloat16().to(dev)) for i, (M, N, K) in enumerate(shapes)]
    widx = [i for i, w in enumerate(is_w) if w]
    hyper = dict(lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, wd=1e-2)

    def problems(grad):
        arr = (capi.GroupedProblem * len(shapes))()
        for i, ((dy, x), (M, N, K)) in enumerate(zip(ops, shapes)):
            arr[i].dy, arr[i].x, arr[i].dw = dy.data_ptr(), x.data_ptr(), grad.data_ptr() + 4 * offs[widx[i]]
            arr[i].M, arr[i].N, arr[i].K = M, N, K
        return arr

    def run(fused):
        p, m, v = p0.clone(), torch.zeros(total, device=dev), torch.zeros(total, device=dev)
        shadow = torch.zeros(total, device=dev, dtype=torch.bfloat16)
        steps = torch.zeros(nparam, device=dev)
        steps[1] = 3.0                                         # one weight matrix is further along (its own bias correction)
        comm = None
        for it in range(2):
            grad = grads_other.clone() * (it + 1)
            steps += 1.0
            if fused:
                arena = capi.AdamWArena(p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), shadow.data_ptr(), steps.data_ptr(), total,
                                        hyper["lr"], hyper["b1"], hyper["b2"], hyper["eps"], hyper["wd"])
                sidx = (ctypes.c_int * len(shapes))(*widx)
                capi.call("unetr_gemm_bf16_grouped_wgrad_adamw", problems(grad), len(shapes), ctypes.byref(arena), sidx, st)
                rows, blocks = [], 0
                for i in range(nparam):
                    if not is_w[i]:
                        rows.append((offs[i], offs[i] + sizes[i], i, blocks))
                        blocks += (sizes[i] + 4095) // 4096
                table = torch.tensor(rows, dtype=torch.int64, device=dev)
                capi.call("unetr_adamw_ranges", ctypes.byref(arena), table.data_ptr(), len(rows), blocks, st)
            else:
                capi.call("unetr_gemm_bf16_grouped_wgrad", problems(grad), len(shapes), st)
                for i in range(nparam):
                    o, n = offs[i], sizes[i]
                    capi.call("unetr_adamw", p.data_ptr() + 4 * o, grad.data_ptr() + 4 * o, m.data_ptr() + 4 * o, v.data_ptr() + 4 * o, n,
                              hyper["lr"], hyper["b1"], hyper["b2"], hyper["eps"], hyper["wd"], steps.data_ptr() + 4 * i,
                              shadow.data_ptr() + 2 * o, st)
            # (b) the communication-buffer form of the same gradients
            comm = torch.full((total,), float("nan"), device=dev, dtype=torch.bfloat16)
            grad2 = grads_other.clone() * (it + 1)
            if fused:
                capi.call("unetr_gemm_bf16_grouped_wgrad_bf16out", problems(grad2), len(shapes), grad2.data_ptr(), comm.data_ptr(), total, st)
                rows, blocks = [], 0
                for i in range(nparam):
                    if not is_w[i]:
                        rows.append((offs[i], offs[i] + sizes[i], blocks))
                        blocks += (sizes[i] + 8191) // 8192
                table = torch.tensor(rows, dtype=torch.int64, device=dev)
                capi.call("unetr_cast_bf16_ranges", grad2.data_ptr(), comm.data_ptr(), table.data_ptr(), len(rows), blocks, st)
            else:
                capi.call("unetr_gemm_bf16_grouped_wgrad", problems(grad2), len(shapes), st)
                capi.call("unetr_cast_bf16", grad2.data_ptr(), comm.data_ptr(), total, st)
        torch.cuda.synchronize()
        return p, m, v, shadow, comm

    ref, fus = run(False), run(True)
    for name, a, b in zip(("param", "exp_avg", "exp_avg_sq", "shadow", "comm buffer"), ref, fus):
        assert not torch.isnan(a.float()).any(), name
        assert torch.equal(a, b), name
    assert (ref[0] != p0).float().mean() > 0.99               # every parameter moved


def test_gemm_bf16_epilogues(pkg, dev):
    Fn = pkg.functional
    M, N, K, L = 432, 512, 256, 216
    x, w = g(M, K, seed=1).bfloat16(), g(N, K, seed=2, scale=0.1).bfloat16()
    b, res, pos, aux = g(N, seed=3), g(M, N, seed=4), g(L, N, seed=5), g(M, N, seed=6)
    xd, wd = x.to(dev), w.to(dev)
    lin = (x.double() @ w.double().t()).float()
    y = torch.empty(M, N, device=dev)
    Fn.gemm_bf16(xd, wd, M, N, K, C=y, bias=b.to(dev), res=res.to(dev), ldr=N)
    assert relerr(y, lin + b + res) < 2e-5
    Fn.gemm_bf16(xd, wd, M, N, K, C=y, bias=b.to(dev), res=pos.to(dev), ldr=N, res_mod=L)
    assert relerr(y, lin + b + pos.repeat(2, 1)) < 2e-5
    pre = torch.empty(M, N, device=dev)
    yb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    Fn.gemm_bf16(xd, wd, M, N, K, C=y, Cb=yb, bias=b.to(dev), act=1, pre=pre)
    assert relerr(pre, lin + b) < 2e-5 and relerr(y, F.gelu(lin + b)) < 2e-5
    assert relerr(yb.float(), F.gelu(lin + b)) < 5e-3
    auxr = aux.clone().requires_grad_(True)
    F.gelu(auxr).sum().backward()
    Fn.gemm_bf16(xd, wd, M, N, K, C=y, act=2, aux=aux.to(dev), ldaux=N)
    assert relerr(y, lin * auxr.grad) < 2e-5
    y0 = g(M, N, seed=7)
    y = y0.to(dev)
    Fn.gemm_bf16(xd, wd, M, N, K, C=y, accumulate=True, alpha=0.5)
    assert relerr(y, y0 + 0.5 * lin) < 2e-5
    assert torch.equal(Fn.cast_bf16(res.to(dev)).cpu(), res.bfloat16())
    odd = g(1003, seed=8)
    assert torch.equal(Fn.cast_bf16(odd.to(dev)).cpu(), odd.bfloat16())


@pytest.mark.parametrize("prec", [0, 1, 2])
def test_gemm_epilogues(pkg, dev, prec):
    Fn = pkg.functional
    M, N, K, L = 432, 512, 256, 216
    x, w, b, res, pos = g(M, K, seed=1), g(N, K, seed=2, scale=0.1), g(N, seed=3), g(M, N, seed=4), g(L, N, seed=5)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    assert relerr(Fn.linear_fwd(xd, wd, bd, prec, res=res.to(dev)), x @ w.t() + b + res) < TOL[prec]
    assert relerr(Fn.linear_fwd(xd, wd, bd, prec, res=pos.to(dev), res_mod=L), x @ w.t() + b + pos.repeat(2, 1)) < TOL[prec]
    pre = torch.empty(M, N, device=dev)
    y = Fn.linear_fwd(xd, wd, bd, prec, act=1, pre=pre)
    u = x @ w.t() + b
    assert relerr(pre, u) < TOL[prec]
    assert relerr(y, F.gelu(u)) < TOL[prec]
    # dgelu epilogue: dx = (dy @ w) * gelu'(aux)
    dy, aux = g(M, N, seed=6), g(M, K, seed=7)
    auxr = aux.clone().requires_grad_(True)
    F.gelu(auxr).backward(dy @ w)
    assert relerr(Fn.linear_dgrad(dy.to(dev), wd, prec, aux=aux.to(dev)), auxr.grad) < TOL[prec]


def test_colsum(pkg, dev):
    Fn = pkg.functional
    for M, N in [(432, 768), (2, 216 * 768), (5000, 37)]:
        x = g(M, N, seed=M)
        assert relerr(Fn.colsum(x.to(dev), M, N, N), x.sum(0)) < 1e-5


@pytest.mark.parametrize("M,H", [(432, 768), (8, 128), (50, 2048), (33, 1024), (19, 772)])   # the 3 / 4 / 8-vector instantiations
def test_layernorm(pkg, dev, M, H):
    Fn = pkg.functional
    x, w, b, dy, dres = g(M, H, seed=1) * 2 + 0.5, g(H, seed=2), g(H, seed=3), g(M, H, seed=4), g(M, H, seed=5)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (H,), wr, br, 1e-5)
    yr.backward(dy)
    y, mean, rstd = Fn.layernorm_fwd(x.to(dev), w.to(dev), b.to(dev))
    assert relerr(y, yr) < 1e-5
    dx, dw, db = Fn.layernorm_bwd(dy.to(dev), x.to(dev), w.to(dev), mean, rstd, dres=dres.to(dev))
    assert relerr(dx, xr.grad + dres) < 1e-5
    assert relerr(dw, wr.grad) < 1e-5
    assert relerr(db, br.grad) < 1e-5


def _attn_ref(qkv, B, L, heads, dh):
    hd = heads * dh
    t = qkv.view(B, L, 3, heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]
    att = (torch.einsum("bhxd,bhyd->bhxy", q, k) * dh ** -0.5).softmax(-1)
    return torch.einsum("bhxy,bhyd->bhxd", att, v).permute(0, 2, 1, 3).reshape(B * L, hd)


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("B,L,heads,dh", [(2, 216, 12, 64), (1, 8, 4, 32), (1, 1000, 2, 64), (2, 100, 3, 128), (1, 33, 2, 32)])
def test_attention(pkg, dev, prec, B, L, heads, dh):
    Fn = pkg.functional
    qkv = g(B * L, 3 * heads * dh, seed=L)
    dout = g(B * L, heads * dh, seed=L + 1)
    qr = qkv.clone().requires_grad_(True)
    ref = _attn_ref(qr, B, L, heads, dh)
    ref.backward(dout)
    out, lse = Fn.attention_fwd(qkv.to(dev), B, L, heads, dh, prec)
    assert relerr(out, ref) < TOL[prec]
    dqkv = Fn.attention_bwd(qkv.to(dev), out, dout.to(dev), lse, B, L, heads, dh, prec)
    hd = heads * dh
    for i, name in enumerate("qkv"):
        assert relerr(dqkv[:, i * hd:(i + 1) * hd], qr.grad[:, i * hd:(i + 1) * hd]) < TOL[prec], name


def cl(x):  # NCDHW -> channels-last
    return x.permute(0, 2, 3, 4, 1).contiguous()


def ncdhw(x):
    return x.permute(0, 4, 1, 2, 3).contiguous()


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("B,S,cin,cout", [(2, 6, 768, 32), (1, 5, 32, 16), (2, 12, 64, 64), (2, 6, 768, 128), (2, 12, 128, 64),
                                          # shapes the dedicated tconv2 kernels take (>= 2048 input voxels, small channel counts):
                                          (2, 16, 32, 16), (1, 17, 16, 8), (1, 13, 64, 32), (1, 16, 48, 24), (2, 12, 32, 32), (1, 14, 16, 64)])
def test_tconv(pkg, dev, prec, B, S, cin, cout):
    Fn = pkg.functional
    x, w, dy = rq(g(B, cin, S, S, S, seed=1), prec), g(cin, cout, 2, 2, 2, seed=2, scale=0.1), rq(g(B, cout, 2 * S, 2 * S, 2 * S, seed=3), prec)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv_transpose3d(xr, wr, stride=2)
    yr.backward(dy)
    dims = (B, S, S, S)
    xd, wd, dyd = act(cl(x), prec, dev), w.to(dev), act(cl(dy), prec, dev)
    y, xb = Fn.tconv_fwd(xd, cin, wd, dims, cin, cout, prec)
    # (bf16 mode, Cin % 64 == 0, few voxels: the bf16-storage GEMM + pixel-shuffle form; xb is the bf16 input it kept)
    assert (xb is not None) == (prec == 1 and cin % 64 == 0 and B * S ** 3 < 8192 and (B * S ** 3) % 8 == 0)
    assert relerr(ncdhw(y.cpu()), yr) < TOL[prec]
    dx, dw = Fn.tconv_bwd(xd, cin, xb, dyd, cout, wd, dims, cin, cout, prec, True)
    assert relerr(ncdhw(dx.cpu()), xr.grad) < TOL[prec]
    assert dw.shape == wr.grad.shape and relerr(dw, wr.grad) < TOL[prec]
    assert relerr(ncdhw(Fn.tconv_dgrad(dyd, cout, wd, dims, cin, cout, prec).cpu()), xr.grad) < TOL[prec]      # the direct kernels
    assert relerr(Fn.tconv_wgrad(xd, cin, dyd, cout, dims, cin, cout, prec), wr.grad) < TOL[prec]
    # write into / read from one half of a concat buffer
    cat = torch.zeros(B, 2 * S, 2 * S, 2 * S, 2 * cout, device=dev, dtype=xd.dtype)
    Fn.tconv_fwd(xd, cin, wd, dims, cin, cout, prec, out=cat, ldo=2 * cout)
    assert relerr(ncdhw(cat[..., :cout].cpu()), yr) < TOL[prec]
    assert cat[..., cout:].abs().max().item() == 0.0


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("B,dims3,cin,cout", [(2, (8, 8, 8), 1, 16), (1, (5, 6, 7), 8, 16), (2, (12, 12, 12), 32, 16), (1, (4, 4, 4), 64, 32),
                                              (1, (6, 5, 4), 4, 16)])
def test_conv3(pkg, dev, prec, B, dims3, cin, cout):
    Fn = pkg.functional
    D, H, W = dims3
    x, w, dy = g(B, cin, D, H, W, seed=1), g(cout, cin, 3, 3, 3, seed=2, scale=0.2), g(B, cout, D, H, W, seed=3)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, padding=1)
    yr.backward(dy)
    dims = (B, D, H, W)
    xd, wd, dyd = cl(x).to(dev), w.to(dev), cl(dy).to(dev)
    y = Fn.conv_fwd(xd, cin, Fn.conv_pack(wd, 0), dims, cin, cout, 3, prec)
    assert relerr(ncdhw(y.cpu()), yr) < TOL[prec]
    dx = Fn.conv_fwd(dyd, cout, Fn.conv_pack(wd, 1), dims, cout, cin, 3, prec)
    assert relerr(ncdhw(dx.cpu()), xr.grad) < TOL[prec]
    assert relerr(Fn.conv_wgrad(xd, cin, dyd, cout, dims, cin, cout, 3, prec), wr.grad) < TOL[prec]


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("B,dims3,cin,cout,with3", [(2, (8, 8, 16), 16, 16, True), (1, (9, 7, 19), 1, 16, True), (2, (6, 10, 20), 32, 16, True),
                                                     (1, (12, 12, 12), 64, 32, True), (1, (5, 6, 7), 16, 32, False), (2, (4, 4, 16), 256, 128, True),
                                                     (1, (10, 9, 33), 4, 16, True), (2, (20, 12, 40), 1, 16, True), (2, (8, 8, 32), 1, 16, False)])
def test_conv3_fused_stats_and_1x1(pkg, dev, prec, B, dims3, cin, cout, with3):
    """Fused residual-block front: conv3x3x3 + InstanceNorm statistics (+ the 1x1x1 conv on the same input with its
    statistics) in one launch == torch conv3d / mean / rstd, over ragged volumes and all kernel modes (pair, slab, scalar)."""
    Fn = pkg.functional
    D, H, W = dims3
    image = cin < 8                       # 1 / 4 channels: the fp32 image in front of encoder1 (stays fp32 in bf16 mode too)
    x = g(B, D, H, W, cin, seed=1) if image else rq(g(B, D, H, W, cin, seed=1), prec)
    w = g(cout, cin, 3, 3, 3, seed=2, scale=0.2)
    w3 = g(cout, cin, 1, 1, 1, seed=3, scale=0.5) if with3 else None
    xn = x.permute(0, 4, 1, 2, 3)
    r = Fn.conv3_fused(x.to(dev) if image else act(x, prec, dev), cin, w.to(dev), w3.to(dev) if with3 else None, (B, D, H, W), prec)
    assert r is not None
    c, st, c3, st3 = r

    def check(out, stats, ref):
        refl = ref.permute(0, 2, 3, 4, 1)
        assert relerr(out, refl) < TOL[prec]
        o = out.cpu().double().reshape(B, -1, cout)              # statistics of what the kernel itself produced
        mu = o.mean(1)
        rstd = 1.0 / torch.sqrt(o.var(1, unbiased=False) + 1e-5)
        # (bf16 mode: the statistics come from the fp32 accumulators, `o` is their bf16-rounded image)
        tol = 1e-4 if prec == 0 else 4e-3
        assert relerr(stats[..., 0], mu.float()) < tol + (mu.abs().max() < 1e-3) and relerr(stats[..., 1], rstd.float()) < tol

    check(c, st, F.conv3d(xn, w, padding=1))
    if with3:
        check(c3, st3, F.conv3d(xn, w3))
    else:
        assert c3 is None and st3 is None


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("B,dims3,cin,cout", [(2, (8, 8, 16), 32, 16), (1, (9, 7, 19), 16, 16), (1, (6, 10, 20), 64, 32), (1, (5, 6, 7), 16, 32),
                                              (2, (4, 4, 16), 256, 128), (1, (10, 9, 33), 48, 16)])
def test_conv3_dgrad_fused(pkg, dev, prec, B, dims3, cin, cout):
    """Input gradient of a residual block in one launch: conv3x3x3^T(dc1; w1) + conv1x1x1^T(dc3; w3) == autograd of
    conv3d(x, w1) and conv3d(x, w3) fed by dc1 / dc3."""
    Fn = pkg.functional
    D, H, W = dims3
    x = g(B, cin, D, H, W, seed=1).requires_grad_(True)
    w1, w3 = g(cout, cin, 3, 3, 3, seed=2, scale=0.2), g(cout, cin, 1, 1, 1, seed=3, scale=0.5)
    dc1, dc3 = rq(g(B, cout, D, H, W, seed=4), prec), rq(g(B, cout, D, H, W, seed=5), prec)
    ((F.conv3d(x, w1, padding=1) * dc1).sum() + (F.conv3d(x, w3) * dc3).sum()).backward()
    dx = torch.empty(B, D, H, W, cin, device=dev, dtype=Fn.act_dtype(prec))
    ok = Fn.conv3_dgrad_fused(act(cl(dc1), prec, dev), act(cl(dc3), prec, dev), w1.to(dev), w3.to(dev), dx, (B, D, H, W), prec)
    assert ok
    assert relerr(ncdhw(dx.cpu()), x.grad) < TOL[prec]


def test_tr16_probe(pkg, dev):
    """ds_read_b64_tr_b16: lane 4q+p of each 16-lane group addresses row q, columns 4p..4p+3 of a 4x16 block of
    16-bit elements; lane i receives column i of the 4 rows (element q = row q)."""
    src = torch.arange(64 * 64, dtype=torch.int32).to(torch.int16)
    out = torch.zeros(64 * 4, dtype=torch.int16, device=dev)
    pkg._capi.call("unetr_debug_tr16", src.to(dev).data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    out = out.cpu().view(64, 4)
    for lane in range(64):
        grp, i = lane >> 4, lane & 15
        for q in range(4):
            assert out[lane, q].item() == (grp * 4 + q) * 64 + i, (lane, q)


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("B,dims3,cin,cout", [(2, (8, 8, 8), 1, 16), (1, (5, 6, 7), 8, 16), (2, (12, 12, 12), 32, 16), (1, (4, 4, 4), 64, 32),
                                              (1, (6, 5, 20), 4, 16), (1, (12, 12, 12), 256, 128), (1, (9, 17, 33), 16, 16), (1, (8, 8, 16), 48, 48)])
def test_conv3_halo(pkg, dev, monkeypatch, prec, B, dims3, cin, cout):
    Fn = pkg.functional
    if cin % 32 == 0 and prec == 1 and B == 1:
        monkeypatch.setenv("UNETR_WG_NSL", "2")        # the two-slabs-per-workgroup weight-gradient form (off by default: slower)
    D, H, W = dims3
    image = cin < 8                       # the fp32 image (1 / 4 channels): fp32 storage in bf16 mode too
    x = g(B, cin, D, H, W, seed=1) if image else rq(g(B, cin, D, H, W, seed=1), prec)
    w, dy = g(cout, cin, 3, 3, 3, seed=2, scale=0.2), rq(g(B, cout, D, H, W, seed=3), prec)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, padding=1)
    yr.backward(dy)
    dims = (B, D, H, W)
    xd, wd, dyd = (cl(x).to(dev) if image else act(cl(x), prec, dev)), w.to(dev), act(cl(dy), prec, dev)
    if not (image and prec == 1):         # (the plain entry point takes feature maps only; the image goes through conv3_fused)
        assert relerr(ncdhw(Fn.conv3(xd, cin, wd, dims, prec).cpu()), yr) < TOL[prec]
    if cin % 16 == 0:
        assert relerr(ncdhw(Fn.conv3(dyd, cout, wd, dims, prec, mode=1).cpu()), xr.grad) < TOL[prec]
        # accumulate into an existing buffer with a wider pitch
        buf = torch.ones(B, D, H, W, 2 * cin, device=dev, dtype=Fn.act_dtype(prec))
        Fn.conv3(dyd, cout, wd, dims, prec, mode=1, out=buf, ldo=2 * cin, accumulate=True)
        assert relerr(ncdhw(buf[..., :cin].cpu()), xr.grad + 1) < TOL[prec]
        assert (buf[..., cin:] == 1).all()
    assert relerr(Fn.conv3_wgrad(xd, cin, dyd, cout, dims, cin, cout, prec), wr.grad) < TOL[prec]


@pytest.mark.parametrize("n,G", [(6912, 512), (432, 1024), (884736, 4), (16, 300), (27 * 48 * 48, 37), (1000, 64), (35, 9)])
def test_reduce_rows_grouped(pkg, dev, n, G):
    """dst[i] = sum over rows of part[row][i], every weight-gradient reduction of a backward pass in one launch: the 16-byte form
    (rows of whole quads) and the scalar form give the same bits (same row phases, same summation tree), both == the fp64 sum."""
    import ctypes
    capi = pkg._capi
    part = g(G, n, seed=1).to(dev)
    outs = []
    for mis in (0, 1):                                  # a destination one float off a 16-byte boundary takes the scalar form
        buf = torch.zeros(n + 8, device=dev)
        dst = buf[mis:mis + n]
        arr = (capi.ReduceProblem * 2)()
        arr[0].part, arr[0].dst, arr[0].n, arr[0].rows = part.data_ptr(), dst.data_ptr(), n, G
        half = torch.zeros(n + 8, device=dev)
        arr[1].part, arr[1].dst, arr[1].n, arr[1].rows = part.data_ptr(), half[mis:mis + n].data_ptr(), n, max(1, G // 2)
        capi.call("unetr_reduce_rows_grouped", arr, 2, torch.cuda.current_stream().cuda_stream)
        outs.append((dst.clone(), half[mis:mis + n].clone()))
        assert float(buf[n + mis:].abs().sum()) == 0.0
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert relerr(outs[0][0], part.double().sum(0).float()) < 1e-5
    assert relerr(outs[0][1], part[:max(1, G // 2)].double().sum(0).float()) < 1e-5


@pytest.mark.parametrize("B,dims3,cin,cout,with3,pitch2,max_wg", [(2, (8, 8, 16), 16, 16, True, False, 0), (1, (9, 7, 19), 32, 16, True, True, 0),
                                                                   (1, (5, 6, 7), 8, 16, False, False, 0), (1, (12, 12, 12), 64, 32, True, False, 0),
                                                                   (2, (4, 4, 16), 256, 128, False, False, 0), (1, (10, 9, 33), 48, 48, True, True, 0),
                                                                   (3, (16, 16, 32), 16, 16, True, False, 4), (1, (6, 5, 20), 20, 12, True, False, 0),
                                                                   (2, (9, 7, 19), 1, 16, True, False, 0), (1, (8, 8, 16), 3, 16, True, False, 0),
                                                                   (1, (8, 8, 16), 4, 16, False, False, 0)])
def test_conv3_wgrad_bf16x3_split_images(pkg, dev, monkeypatch, B, dims3, cin, cout, with3, pitch2, max_wg):
    """bf16x3 weight gradient on (hi, lo) bf16 images and transposing reads (csrc/conv3.hip: conv3_wgrad_x3_kernel): 3x3x3 and the
    1x1x1 branch's gradient from the same pass, ragged volumes, 8 / 20 / 48 channels (masked slabs), the image block's 1 / 3 / 4 input
    channels (scalar window loads), x read through a wider pitch (the concatenation buffer), a workgroup walking many tiles -- against torch fp32, and against the 4-byte fragment path it replaces
    (UNETR_X3_WGRAD_TR16=0)."""
    Fn = pkg.functional
    D, H, W = dims3
    if max_wg:
        monkeypatch.setenv("UNETR_TEST_MAX_WG", str(max_wg))
    x, dy, dy3 = g(B, cin, D, H, W, seed=1), g(B, cout, D, H, W, seed=2), g(B, cout, D, H, W, seed=3)
    w, w3 = g(cout, cin, 3, 3, 3, seed=4, scale=0.2).requires_grad_(True), g(cout, cin, 1, 1, 1, seed=5).requires_grad_(True)
    xr = x.clone()
    F.conv3d(xr, w, padding=1).backward(dy)
    if with3:
        F.conv3d(xr, w3).backward(dy3)
    ldx = 2 * cin if pitch2 else cin
    xd = torch.zeros(B, D, H, W, ldx, device=dev)
    xd[..., :cin] = cl(x).to(dev)
    dyd, dy3d = cl(dy).to(dev), cl(dy3).to(dev)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("UNETR_X3_WGRAD_TR16", mode)
        out3 = torch.empty(cout, cin, 1, 1, 1, device=dev) if with3 else None
        dw = Fn.conv3_wgrad(xd, ldx, dyd, cout, (B, D, H, W), cin, cout, 2, dy3=dy3d if with3 else None, out3=out3)
        res[mode] = (dw.cpu(), out3.cpu() if with3 else None)
    for mode in ("1", "0"):
        assert relerr(res[mode][0], w.grad) < TOL[2]
        if with3:
            assert relerr(res[mode][1], w3.grad) < TOL[2]
    assert relerr(res["1"][0], res["0"][0]) < TOL[2]


@pytest.mark.parametrize("B,dims3,max_wg", [(2, (9, 7, 19), 0), (2, (20, 12, 40), 0), (3, (16, 16, 32), 8)])
def test_conv3_single_channel_image(pkg, dev, monkeypatch, B, dims3, max_wg):
    """The dedicated kernels for the 1-channel image in front of encoder1 (UnetrBasicBlock(in_channels=1, feature_size=16),
    unetr.py:90-98) in bf16 mode -- forward: 27 taps as the K dimension of ONE MFMA per 16 voxels (+ the 1x1x1 branch on the VALU,
    + both InstanceNorm statistics); weight gradient: taps as the output column -- against torch, over ragged volumes, several
    batch items, and (max_wg) workgroups that walk many tiles across batch items.  The generic pair-mode kernels
    (UNETR_CONV_C1_OFF=1) must give the same results up to accumulation order."""
    Fn = pkg.functional
    if max_wg:
        monkeypatch.setenv("UNETR_TEST_MAX_WG", str(max_wg))
    D, H, W = dims3
    x = g(B, 1, D, H, W, seed=1)
    w, w3 = g(16, 1, 3, 3, 3, seed=2, scale=0.2), g(16, 1, 1, 1, 1, seed=3, scale=0.5)
    dy, dy3 = rq(g(B, 16, D, H, W, seed=4), 1), rq(g(B, 16, D, H, W, seed=5), 1)
    xb = x.bfloat16().float()                       # what the kernels contract: bf16-rounded image and weights
    wr, w3r = w.bfloat16().float().requires_grad_(True), w3.bfloat16().float().requires_grad_(True)
    yr, y3r = F.conv3d(xb, wr, padding=1), F.conv3d(xb, w3r)
    (yr * dy).sum().backward()
    (y3r * dy3).sum().backward()
    xd, dims = cl(x).to(dev), (B, D, H, W)
    res = {}
    for off in ("", "1"):
        if off:
            monkeypatch.setenv("UNETR_CONV_C1_OFF", "1")
        c, st, c3, st3 = Fn.conv3_fused(xd, 1, w.to(dev), w3.to(dev), dims, 1)
        assert relerr(ncdhw(c.float().cpu()), yr) < 5e-3 and relerr(ncdhw(c3.float().cpu()), y3r) < 5e-3
        for out, stats in ((c, st), (c3, st3)):
            o = out.float().cpu().double().reshape(B, -1, 16)
            assert relerr(stats[..., 0], o.mean(1).float()) < 4e-3
            assert relerr(stats[..., 1], (1.0 / torch.sqrt(o.var(1, unbiased=False) + 1e-5)).float()) < 4e-3
        dw3 = torch.empty(16, 1, 1, 1, 1, device=dev)
        dw = Fn.conv3_wgrad(xd, 1, act(cl(dy), 1, dev), 16, dims, 1, 16, 1, dy3=act(cl(dy3), 1, dev), out3=dw3)
        assert relerr(dw, wr.grad) < 2e-3 and relerr(dw3, w3r.grad) < 2e-3
        res[off] = (c.float(), c3.float(), st, st3, dw.clone(), dw3.clone())
    for a, b in zip(res[""], res["1"]):
        assert relerr(a, b) < 2e-3
    assert torch.equal(res[""][1], res["1"][1])      # the 1x1x1 branch is the exact product of two bf16 numbers on both paths


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("B,S,C", [(2, 12, 16), (1, 20, 32), (2, 6, 128)])
def test_instnorm(pkg, dev, prec, B, S, C):
    """prec 1 = bf16-stored feature maps (fp32 statistics and arithmetic): compared on the bf16-rounded inputs, the outputs
    carry one bf16 rounding"""
    Fn = pkg.functional
    V = S ** 3
    x, x2, dy = rq(g(B, C, S, S, S, seed=1) * 1.5 + 0.3, prec), rq(g(B, C, S, S, S, seed=2) * 0.7 - 0.2, prec), rq(g(B, C, S, S, S, seed=3), prec)
    xd, x2d, dyd = act(cl(x), prec, dev), act(cl(x2), prec, dev), act(cl(dy), prec, dev)
    t1, t2 = (1e-5, 2e-5) if prec == 0 else (5e-3, 5e-3)
    # single branch with lrelu
    xr = x.clone().requires_grad_(True)
    yr = F.leaky_relu(F.instance_norm(xr, eps=1e-5), 0.01)
    yr.backward(dy)
    sa = Fn.instnorm_stats(xd, C, B, V, C)
    assert relerr(ncdhw(Fn.instnorm_apply(xd, sa, B, V, C, True).cpu()), yr) < t1
    dx, _ = Fn.instnorm_bwd(dyd, C, xd, sa, B, V, C, True)
    assert relerr(ncdhw(dx.cpu()), xr.grad) < t2
    # two branches + lrelu
    xr, x2r = x.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    yr = F.leaky_relu(F.instance_norm(xr, eps=1e-5) + F.instance_norm(x2r, eps=1e-5), 0.01)
    yr.backward(dy)
    sb = Fn.instnorm_stats(x2d, C, B, V, C)
    assert relerr(ncdhw(Fn.instnorm_apply(xd, sa, B, V, C, True, x2=x2d, sb=sb).cpu()), yr) < t1
    dx, dx2 = Fn.instnorm_bwd(dyd, C, xd, sa, B, V, C, True, x2=x2d, sb=sb)
    assert relerr(ncdhw(dx.cpu()), xr.grad) < t2
    assert relerr(ncdhw(dx2.cpu()), x2r.grad) < t2


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("cin,cout", [(16, 4), (16, 14), (32, 3), (16, 2)])
def test_outconv_and_layout(pkg, dev, prec, cin, cout):
    """UnetOutBlock forward + backward; (16, 14) is the reference's default head (14 BTCV classes,
    unetr_segmentation_3d.py:303): its weight gradient takes the generic kernel on fp32- and bf16-stored feature maps"""
    Fn = pkg.functional
    B, S = 2, 10
    x, w, b, dl = rq(g(B, cin, S, S, S, seed=1), prec), g(cout, cin, 1, 1, 1, seed=2), g(cout, seed=3), g(B, cout, S, S, S, seed=4)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, br)
    yr.backward(dl)
    xd = act(cl(x), prec, dev).requires_grad_(True)
    wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = Fn.OutConvFn.apply(xd, wd, bd)
    assert y.dtype == torch.float32 and relerr(y, yr) < 1e-5
    y.backward(dl.to(dev))
    assert xd.grad.dtype == xd.dtype and relerr(ncdhw(xd.grad.cpu()), xr.grad) < (1e-5 if prec == 0 else 5e-3)
    assert relerr(wd.grad, wr.grad) < 1e-4
    assert relerr(bd.grad, br.grad) < 1e-4
    # layout moves
    t = act(cl(x), prec, dev).requires_grad_(True)
    n = Fn.ToNCDHWFn.apply(t)
    assert n.dtype == torch.float32 and torch.equal(n.cpu(), x)
    n.backward(x.to(dev) * 2)
    assert t.grad.dtype == t.dtype and torch.equal(t.grad.float().cpu(), cl(x) * 2)
    x4 = g(2, 4, 6, 5, 7, seed=9)
    assert torch.equal(Fn.to_channels_last(x4.to(dev)).cpu(), cl(x4))


def test_patch_embed(pkg, dev):
    Fn = pkg.functional
    from oracle.unetr_oracle import _PerceptronPatches
    for C in (1, 2):
        x = g(2, C, 32, 32, 48, seed=C)
        hid, pd = 64, C * 4096
        L = 2 * 2 * 3
        w, b, pos = g(hid, pd, seed=5, scale=0.02), g(hid, seed=6), g(1, L, hid, seed=7)
        wr, br, posr = w.clone().requires_grad_(True), b.clone().requires_grad_(True), pos.clone().requires_grad_(True)
        ref = F.linear(_PerceptronPatches((16, 16, 16))(x), wr, br) + posr
        dz = g(2, L, hid, seed=8)
        ref.backward(dz)
        wd, bd, posd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True), pos.to(dev).requires_grad_(True)
        z = Fn.PatchEmbedFn.apply(x.to(dev), wd, bd, posd, 16, 0)
        assert relerr(z, ref.reshape(2 * L, hid)) < 2e-5
        z.backward(dz.reshape(2 * L, hid).to(dev))
        assert relerr(wd.grad, wr.grad) < 2e-5
        assert relerr(bd.grad, br.grad) < 2e-5
        assert relerr(posd.grad, posr.grad) < 2e-5


@pytest.mark.parametrize("B,C,S", [(2, 4, 24), (1, 2, 17), (2, 3, 32)])
def test_dicece(pkg, dev, B, C, S):
    from oracle.unetr_oracle import oracle_dice_ce_terms
    logits = g(B, C, S, S, S, seed=1) * 2
    label = torch.randint(0, C, (B, 1, S, S, S), generator=torch.Generator().manual_seed(2)).float()
    lr = logits.clone().requires_grad_(True)
    d, c = oracle_dice_ce_terms(lr, label)
    (d + c).backward()
    ld = logits.to(dev).requires_grad_(True)
    crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)
    t = crit.terms(ld, label.to(dev))
    assert relerr(t[1], d) < 1e-5 and relerr(t[2], c) < 1e-5 and relerr(t[0], d + c) < 1e-5
    (t[0] * 1.0).backward()
    assert relerr(ld.grad, lr.grad) < 1e-4


@pytest.mark.parametrize("B,C,S", [(2, 3, 24), (1, 4, 17), (2, 2, 32)])
def test_dicece_sigmoid_multilabel(pkg, dev, B, C, S):
    """DiceCELoss(to_onehot_y=False, sigmoid=True) on an overlapping multi-label mask (unetr_segmentation_3d.py:477-482):
    Dice on sigmoid probabilities, CE against argmax_c(target) -- including all-zero voxels (argmax -> channel 0)."""
    from oracle.unetr_oracle import oracle_dice_ce_terms
    logits = g(B, C, S, S, S, seed=1) * 2
    gen = torch.Generator().manual_seed(2)
    target = (torch.rand(B, C, S, S, S, generator=gen) < 0.35).float()      # channels overlap, many voxels are all-zero
    lr = logits.clone().requires_grad_(True)
    d, c = oracle_dice_ce_terms(lr, target, to_onehot_y=False, softmax=False, sigmoid=True)
    (d + c).backward()
    ld = logits.to(dev).requires_grad_(True)
    crit = pkg.DiceCELoss(to_onehot_y=False, sigmoid=True)
    t = crit.terms(ld, target.to(dev))
    assert relerr(t[1], d) < 1e-5 and relerr(t[2], c) < 1e-5 and relerr(t[0], d + c) < 1e-5
    (t[0] * 1.0).backward()
    assert relerr(ld.grad, lr.grad) < 1e-4
    with pytest.raises(ValueError):
        crit(ld, target[:, :1].to(dev))


def test_adamw(pkg, dev):
    n = 100003
    p, gr = g(n, seed=1), g(n, seed=2)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-3, weight_decay=1e-2)
    pd = torch.zeros(n + 1, device=dev)[:n]  # 16B-aligned base
    pd.copy_(p)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    shadow = torch.zeros(n, device=dev, dtype=torch.bfloat16)
    step = torch.zeros(1, device=dev)
    for it in range(3):
        pr.grad = gr * (it + 1)
        opt.step()
        step += 1
        gd = (gr * (it + 1)).to(dev)
        pkg._capi.call("unetr_adamw", pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, 1e-2,
                       step.data_ptr(), shadow.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert relerr(pd, pr) < 1e-5
    assert torch.equal(shadow.cpu(), pd.cpu().bfloat16())      # bf16 weight shadow written by the same kernel


@pytest.mark.parametrize("kind", ["ranking", "contrastive"])
@pytest.mark.parametrize("shape,slice_dim,init_idx", [((4, 8, 12, 12, 12), 2, 1), ((4, 8, 12, 12, 12), 3, 0), ((4, 8, 12, 12, 12), 4, 2),
                                                      ((4, 3, 16, 8, 24), 3, 1), ((4, 128, 12, 12, 12), 2, 2), ((4, 2, 24, 24, 24), 4, 5)])
def test_ranking_losses(pkg, dev, kind, shape, slice_dim, init_idx):
    """Fused BT / contrastive loss vs the line-by-line restatement of unetr_ranking_pretraining_3d.py:59-133,202-236
    (this part of the reference is in-repo code, so its parity is pinned by the reference itself)."""
    from oracle.unetr_oracle import oracle_bt_loss, oracle_contrastive_loss, oracle_extract_triplets
    if kind == "contrastive" and shape[1] * shape[2] * shape[3] * shape[4] > 8 * 12 ** 3:
        pytest.skip("the CPU oracle's 576 x 577-term Python loop takes minutes at this size: these two sizes are held to outputs of "
                    "the REFERENCE's own functions instead (test_ranking_losses_large_vs_reference_fixture)")
    T = 0.1 if kind == "ranking" else 0.5
    feat = g(*shape, seed=3) + 0.3
    fr = feat.clone().double().requires_grad_(True)
    f1, f2 = torch.split(fr, [2, 2], dim=0)
    r, s, d = oracle_extract_triplets(f1, f2, slice_dim, init_idx)
    ref = (oracle_bt_loss if kind == "ranking" else oracle_contrastive_loss)(r, s, d, T)
    ref.backward()
    fd = feat.to(dev).requires_grad_(True)
    loss = pkg.ranking_loss(fd, slice_dim, init_idx, T, kind=kind)
    assert relerr(loss, ref) < 2e-5
    (loss * 2.0).backward()
    assert relerr(fd.grad, 2.0 * fr.grad) < 2e-4


@pytest.mark.gpu
def test_ranking_losses_vs_reference_fixture(pkg, dev):
    """The HIP ranking kernels against outputs of the REFERENCE's own code (tests/golden/ranking_ref.npz, written by
    tests/golden/make_ranking_golden.py from unetr_ranking_pretraining_3d.py:59-133,202-236): all three slice axes, both
    loss kinds, two feature shapes; loss within 2e-5 relative, input gradient within 2e-4 of its max (fp32 kernels vs
    the reference's fp64 run; the reference's own fp32 run differs from its fp64 run by up to 2e-6 / loss)."""
    from test_oracle_cpu import ranking_cases
    n = 0
    for key, feat, axis, kind, init_idx, T, loss64, loss32, grad in ranking_cases():
        fd = feat.to(dev).requires_grad_(True)
        loss = pkg.ranking_loss(fd, axis, init_idx, T, kind=kind)
        assert abs(loss.item() - loss64) <= 2e-5 * abs(loss64), (key, loss.item(), loss64)
        loss.backward()
        assert relerr(fd.grad, grad) < 2e-4, key
        n += 1
    assert n == 12


@pytest.mark.gpu
def test_ranking_losses_large_vs_reference_fixture(pkg, dev):
    """BASELINE config[4]'s two feature sizes at 96^3 -- enc4 [4,128,12,12,12] and the logits-like [4,2,24,24,24] -- against
    tests/golden/ranking_ref_large.npz: outputs of the reference's own BTLoss / ContrastiveLoss
    (unetr_ranking_pretraining_3d.py:202-236) executed by tests/golden/make_ranking_golden.py.  BT: loss and every 97th
    element of the input gradient; contrastive: the loss (its autograd graph does not fit the build host at these sizes)."""
    import os
    import numpy as np
    import importlib.util
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(here, "ranking_ref_large.npz"))
    spec = importlib.util.spec_from_file_location("make_ranking_golden", os.path.join(here, "make_ranking_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)                     # (only its deterministic input generator `features` is used)
    T, stride = float(z["temperature"]), int(z["grad_stride"])
    for i, (C, S, axis, contrastive, seed) in enumerate(z["cases"].tolist()):
        feat = gen.features(C, S, seed, torch.float32)
        fd = feat.to(dev).requires_grad_(True)
        loss = pkg.ranking_loss(fd, axis, int(z[f"c{i}_init_idx"]), T, kind="contrastive" if contrastive else "ranking")
        ref = float(z[f"c{i}_loss"])
        assert abs(loss.item() - ref) <= 5e-5 * abs(ref), (i, loss.item(), ref)
        loss.backward()
        assert torch.isfinite(fd.grad).all()
        if f"c{i}_grad_sub" in z.files:
            got = fd.grad.flatten()[::stride].cpu().numpy()
            assert np.abs(got - z[f"c{i}_grad_sub"]).max() <= 3e-4 * float(z[f"c{i}_grad_absmax"]), i


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(37, 50, 44), (432, 768, 770), (70, 33, 4100), (16, 16, 70001)])
def test_gemm_padding_lanes_ignore_inf_nan(pkg, dev, prec, M, N, K):
    """Ragged M / N / K: the loaders fetch out-of-range lanes from a clamped address (element (0,0), row 0 or column 0 of
    the operand) and must mask the BITS -- an Inf / NaN sitting there may only affect the outputs that really use it."""
    Fn = pkg.functional
    x, w, dy = g(M, K, seed=1), g(N, K, seed=2), g(M, N, seed=3)
    x[0, 0], w[0, 0], dy[0, 0] = float("inf"), float("nan"), float("-inf")
    xd, wd, dyd = x.to(dev), w.to(dev), dy.to(dev)
    y = Fn.linear_fwd(xd, wd, None, prec).cpu()
    ref = x @ w.t()
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isfinite(y), fin)                       # row 0 and column 0 are poisoned, nothing else
    assert fin[1:, 1:].all() and relerr(y[1:, 1:], ref[1:, 1:]) < TOL[prec]
    dx = Fn.linear_dgrad(dyd, wd, prec).cpu()                         # dx = dy @ w: row 0 (dy) and column 0 (w[0,0] hits k=0 only)
    rdx = dy @ w
    assert torch.equal(torch.isfinite(dx), torch.isfinite(rdx))
    ok = torch.isfinite(rdx)
    assert relerr(dx[ok], rdx[ok]) < TOL[prec]
    dw = Fn.linear_wgrad(dyd, xd, prec).cpu()                         # dw = dy^T x
    rdw = dy.t() @ x
    assert torch.equal(torch.isfinite(dw), torch.isfinite(rdw))
    ok = torch.isfinite(rdw)
    assert relerr(dw[ok], rdw[ok]) < TOL[prec]


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,act", [(432, 2304, 768, 0), (432, 3072, 768, 1), (16, 384, 128, 0), (8, 512, 128, 1), (1000, 576, 192, 0),
                                       (70, 200, 256, 1), (432, 768, 1024, 0)])
def test_ln_gemm_bf16(pkg, dev, M, N, K, act):
    """LayerNorm fused as the GEMM prologue (csrc/encoder.hip) vs torch: LayerNorm in fp32, rows rounded to bf16, fp64 product
    with the bf16 weights, bias, exact GELU.  Also the by-products backward needs: normalised rows, mean, rstd, pre-activation."""
    Fn = pkg.functional
    x = g(M, K, seed=1) * 1.7 + 0.3
    gam, bet = 1.0 + 0.2 * g(K, seed=2), 0.1 * g(K, seed=3)
    w = (g(N, K, seed=4) * 0.05).bfloat16()
    bias = 0.1 * g(N, seed=5)
    xn_ref = F.layer_norm(x, (K,), gam, bet, 1e-5)
    xnb = xn_ref.bfloat16()
    pre_ref = (xnb.double() @ w.double().t() + bias.double()).float()
    out_ref = F.gelu(pre_ref) if act else pre_ref
    xd = x.to(dev)
    C = torch.empty(M, N, device=dev)
    Cb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    pre = torch.empty(M, N, device=dev)
    xn = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    Fn.ln_gemm_bf16(xd, gam.to(dev), bet.to(dev), w.to(dev), bias=bias.to(dev), act=act, C=C, Cb=Cb, pre=pre, xn=xn, mean=mean, rstd=rstd)
    # the kernel's bf16 rounding of the normalised rows may differ from torch's by one ulp where fp32 LN differs in the last bit
    assert (xn.float().cpu() - xnb.float()).abs().max() <= 2 ** -7 * xnb.float().abs().max()
    assert relerr(mean, x.mean(1)) < 1e-5 and relerr(rstd, (x.var(1, unbiased=False) + 1e-5).rsqrt()) < 1e-5
    pre_own = (xn.float().cpu().double() @ w.double().t() + bias.double()).float()     # product of the rows the kernel itself kept
    assert relerr(pre, pre_own) < 2e-5
    assert relerr(pre, pre_ref) < 5e-3
    own = F.gelu(pre_own) if act else pre_own
    assert relerr(C, own) < 2e-5
    assert torch.equal(Cb.cpu(), C.cpu().bfloat16())
    # outputs optional: bf16 only, nothing kept
    Cb2 = torch.empty_like(Cb)
    Fn.ln_gemm_bf16(xd, gam.to(dev), bet.to(dev), w.to(dev), bias=bias.to(dev), act=act, Cb=Cb2)
    assert torch.equal(Cb2, Cb)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["6464", "6432", "3264", "64128", "6496"])
def test_gemm_bf16_small_m_tiles(pkg, dev, cfg, monkeypatch):
    """every small-M tile shape of the bf16-storage GEMM (normally chosen by workgroup count) on ragged shapes, both B layouts"""
    Fn = pkg.functional
    monkeypatch.setenv("UNETR_GEMM_CFG", cfg)
    for (M, N, K) in ((432, 768, 768), (100, 200, 192), (432, 3072, 768), (33, 136, 64)):
        x, w, dy = g(M, K, seed=1).bfloat16(), g(N, K, seed=2).bfloat16(), g(M, N, seed=3).bfloat16()
        y = torch.empty(M, N, device=dev)
        Fn.gemm_bf16(x.to(dev), w.to(dev), M, N, K, C=y)
        assert relerr(y, (x.double() @ w.double().t()).float()) < 2e-5, (cfg, M, N, K)
        if N % 64 == 0 and cfg not in ("6432", "6496"):
            dx = torch.empty(M, K, device=dev)
            Fn.gemm_bf16(dy.to(dev), w.to(dev), M, K, N, b_kn=True, C=dx)
            assert relerr(dx, (dy.double() @ w.double()).float()) < 2e-5, (cfg, M, N, K, "b_kn")


@pytest.mark.gpu
@pytest.mark.parametrize("nw", [2, 4, 8])
@pytest.mark.parametrize("B,L,heads", [(2, 216, 12), (1, 1000, 3), (3, 230, 1), (1, 40, 2)])
def test_attention_bf16_forward_wave_counts(pkg, dev, monkeypatch, nw, B, L, heads):
    """the forward attention kernel with 32 / 64 / 128 queries per workgroup (UNETR_ATTN_NW; 128 = the form deep grids -- batch
    >= 16 at 216 tokens -- take by themselves): same output, same log-sum-exp"""
    monkeypatch.setenv("UNETR_ATTN_NW", str(nw))
    Fn = pkg.functional
    dh, Hd = 64, heads * 64
    qkv = (g(B * L, 3 * Hd, seed=1) * 0.8).bfloat16()
    t = qkv.float().view(B, L, 3, heads, dh).permute(2, 0, 3, 1, 4)
    sc = t[0] @ t[1].transpose(-1, -2) * dh ** -0.5
    out_ref = (torch.softmax(sc, dim=-1) @ t[2]).permute(0, 2, 1, 3).reshape(B * L, Hd)
    outb = torch.empty(B * L, Hd, device=dev, dtype=torch.bfloat16)
    out = torch.empty(B * L, Hd, device=dev)
    lse = Fn.attention_bf16_fwd(qkv.to(dev), B, L, heads, dh, outb, out=out)
    assert relerr(out, out_ref) < 1e-2 and torch.equal(outb.cpu(), out.cpu().bfloat16())
    assert relerr(lse, torch.logsumexp(sc, dim=-1)) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("B,L,heads", [(2, 216, 12), (1, 8, 2), (1, 1000, 3), (3, 230, 1), (1, 512, 2)])
def test_attention_bf16_storage(pkg, dev, B, L, heads):
    """attention on bf16-stored q/k/v (csrc/attention_b16.hip: LDS-DMA staged images, one image for row and transposed
    reads, chunk-resident softmax; L = 1000 / 230 / 512 exercise the chunk loop with a ragged last chunk) vs torch on the
    same bf16-rounded inputs; bf16 rounding of P / dS bounds the error."""
    Fn = pkg.functional
    dh, Hd = 64, heads * 64
    qkv = (g(B * L, 3 * Hd, seed=1) * 0.8).bfloat16()
    dout = (g(B * L, Hd, seed=2)).bfloat16()
    ref_in = qkv.float().requires_grad_(True)
    t = ref_in.view(B, L, 3, heads, dh).permute(2, 0, 3, 1, 4)          # "b l (qkv h d) -> qkv b h l d"
    att = torch.softmax(t[0] @ t[1].transpose(-1, -2) * dh ** -0.5, dim=-1)
    out_ref = (att @ t[2]).permute(0, 2, 1, 3).reshape(B * L, Hd)
    out_ref.backward(dout.float())
    qd = qkv.to(dev)
    outb = torch.empty(B * L, Hd, device=dev, dtype=torch.bfloat16)
    out = torch.empty(B * L, Hd, device=dev)
    lse = Fn.attention_bf16_fwd(qd, B, L, heads, dh, outb, out=out)
    assert relerr(out, out_ref) < 1e-2
    assert torch.equal(outb.cpu(), out.cpu().bfloat16())
    lse_ref = torch.logsumexp(t[0] @ t[1].transpose(-1, -2) * dh ** -0.5, dim=-1)
    assert relerr(lse, lse_ref) < 1e-4
    dq32 = torch.empty(B * L, 3 * Hd, device=dev)
    dqb = Fn.attention_bf16_bwd(qd, outb, dout.to(dev), lse, B, L, heads, dh, dqkv=dq32)
    assert relerr(dq32, ref_in.grad) < 2e-2
    assert torch.equal(dqb.cpu(), dq32.cpu().bfloat16())


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("producer", ["tconv", "resblock"])
def test_skip_written_into_concat_buffer(pkg, dev, prec, producer):
    """torch.cat((up, skip), dim=1) of UnetrUpBlock (unetr.py:135-174) without the copy: the skip's producer writes the second
    half of the concatenation buffer (to_cat), UpBlockFn(skip_in_cat=True) the first.  Bit-identical to the copying path,
    outputs and every gradient."""
    Fn = pkg.functional
    B, S, C = 2, 8, 16
    adt = Fn.act_dtype(prec)
    inp = act(cl(g(B, 2 * C, S, S, S, seed=1)), prec, dev)                       # decoder input at half resolution
    wt = (g(2 * C, C, 2, 2, 2, seed=2) * 0.2).to(dev)
    w1, w2, w3 = (g(C, 2 * C, 3, 3, 3, seed=3) * 0.1).to(dev), (g(C, C, 3, 3, 3, seed=4) * 0.1).to(dev), (g(C, 2 * C, 1, 1, 1, seed=5) * 0.2).to(dev)
    if producer == "tconv":
        src = act(cl(g(B, 32, S, S, S, seed=6)), prec, dev)
        pw = [(g(32, C, 2, 2, 2, seed=7) * 0.2).to(dev)]
        make = lambda s, p, to_cat: Fn.TconvFn.apply(s, p[0], prec, to_cat)
    else:
        src = act(cl(g(B, 8, 2 * S, 2 * S, 2 * S, seed=6)), prec, dev)
        pw = [(g(C, 8, 3, 3, 3, seed=7) * 0.1).to(dev), (g(C, C, 3, 3, 3, seed=8) * 0.1).to(dev), (g(C, 8, 1, 1, 1, seed=9) * 0.2).to(dev)]
        make = lambda s, p, to_cat: Fn.ResBlockFn.apply(s, *p, prec, to_cat)
    dout = act(cl(g(B, C, 2 * S, 2 * S, 2 * S, seed=10)), prec, dev)
    res = []
    for in_cat in (False, True):
        leaves = [t.clone().requires_grad_(True) for t in (inp.float(), src.float(), wt, w1, w2, w3, *pw)]
        i_, s_, wt_, w1_, w2_, w3_, *pw_ = leaves
        skip = make(s_.to(adt), pw_, in_cat)
        assert skip.shape == (B, 2 * S, 2 * S, 2 * S, C)
        assert (skip.stride(-2) == 2 * C) == in_cat
        out = Fn.UpBlockFn.apply(i_.to(adt), skip, wt_, w1_, w2_, w3_, prec, in_cat)
        out.backward(dout)
        res.append([out.detach()] + [t.grad for t in leaves])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # a tensor that is not the second half of such a buffer is refused rather than overwritten
    with pytest.raises(RuntimeError):
        Fn.UpBlockFn.apply(inp, torch.zeros(B, 2 * S, 2 * S, 2 * S, C, device=dev, dtype=adt), wt, w1, w2, w3, prec, True)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("B,S,C,ncls", [(2, 8, 16, 4), (1, 6, 16, 2), (2, 5, 16, 3), (1, 8, 8, 4), (2, 6, 16, 14)])
def test_upblock_with_out_conv_head(pkg, dev, monkeypatch, prec, B, S, C, ncls):
    """decoder2 + UnetOutBlock (unetr.py:165-175,206-207) as ONE Function: the block end lrelu(IN(c2) + IN(c3)) is formed inside the
    out conv's kernels and never stored (csrc/norm_misc.hip: outconv_in_fwd_kernel / outconv_in_bwd_kernel).  Fused against the
    stored sequence (UNETR_AMD_OUT_FUSE=0) to rounding -- logits and all eight gradients -- and both against torch autograd;
    14 classes and 8 channels (bf16: one piece per voxel) also run: the first declines the fused kernels (Cout > 4) and must take
    the stored sequence by itself."""
    Fn = pkg.functional
    adt = Fn.act_dtype(prec)
    inp0 = rq(g(B, 2 * C, S, S, S, seed=1), prec)
    skip0 = rq(g(B, C, 2 * S, 2 * S, 2 * S, seed=2), prec)
    ws0 = [g(2 * C, C, 2, 2, 2, seed=3) * 0.2, g(C, 2 * C, 3, 3, 3, seed=4) * 0.1, g(C, C, 3, 3, 3, seed=5) * 0.1, g(C, 2 * C, 1, 1, 1, seed=6) * 0.2,
           g(ncls, C, 1, 1, 1, seed=7) * 0.3, g(ncls, seed=8)]
    dl = g(B, ncls, 2 * S, 2 * S, 2 * S, seed=9)
    # torch autograd on the same (storage-rounded) inputs
    ir, sr = inp0.clone().requires_grad_(True), skip0.clone().requires_grad_(True)
    wr = [t.clone().requires_grad_(True) for t in ws0]
    cat = torch.cat((F.conv_transpose3d(ir, wr[0], stride=2), sr), dim=1)
    a1 = F.leaky_relu(F.instance_norm(F.conv3d(cat, wr[1], padding=1)), 0.01)
    blk = F.leaky_relu(F.instance_norm(F.conv3d(a1, wr[2], padding=1)) + F.instance_norm(F.conv3d(cat, wr[3])), 0.01)
    ref = F.conv3d(blk, wr[4], wr[5])
    ref.backward(dl)
    ref_all = [ref.detach(), cl(ir.grad), cl(sr.grad)] + [t.grad for t in wr]
    res = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("UNETR_AMD_OUT_FUSE", fuse)
        leaves = [cl(inp0).to(dev).requires_grad_(True), cl(skip0).to(dev).requires_grad_(True)] + [t.to(dev).requires_grad_(True) for t in ws0]
        i_, s_, wt, w1, w2, w3, wo, bo = leaves
        logits = Fn.UpBlockFn.apply(i_.to(adt), s_.to(adt), wt, w1, w2, w3, prec, False, wo, bo)
        assert logits.shape == (B, ncls, 2 * S, 2 * S, 2 * S) and logits.dtype == torch.float32
        logits.backward(dl.to(dev))
        res[fuse] = [logits.detach().cpu()] + [t.grad.float().cpu() for t in leaves]
    tol_pair = 2e-4 if prec != 1 else 2e-2
    tol_ref = 2e-3 if prec != 1 else 6e-2
    for k, (a, b) in enumerate(zip(res["0"], res["1"])):
        assert relerr(a, b) < tol_pair, k
    for k, (a, r) in enumerate(zip(res["1"], ref_all)):
        if k > 0:       # gradients by direction: a LeakyReLU mask that flips on a rounding difference moves single elements by O(1)
            assert F.cosine_similarity(a.double().flatten(), r.double().flatten(), dim=0).item() > (0.995 if prec == 1 else 0.9995), k
        else:
            assert relerr(a, r) < tol_ref, k


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,dims3", [(16, 16, (36, 32, 64)), (32, 16, (36, 32, 64)), (64, 32, (36, 32, 64))])
def test_conv3_long_tile_walks(pkg, dev, monkeypatch, cin, cout, dims3):
    """A persistent workgroup that walks MORE than 64 tiles (the per-workgroup tile table is refilled every 64; the bench
    shapes stay below that): the grid is capped at 8 workgroups through the test hook, so each walks 72 tiles.
    Forward (pair layout with the LDS-DMA window / slab layout with one and two weight images), data gradient, weight
    gradient, in bf16 mode against torch on the bf16-rounded operands."""
    Fn = pkg.functional
    monkeypatch.setenv("UNETR_TEST_MAX_WG", "8")
    B, prec = 2, 1
    D, H, W = dims3
    x, w, dy = rq(g(B, cin, D, H, W, seed=1), prec), g(cout, cin, 3, 3, 3, seed=2, scale=0.2), rq(g(B, cout, D, H, W, seed=3), prec)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, padding=1)
    yr.backward(dy)
    dims = (B, D, H, W)
    xd, wd, dyd = act(cl(x), prec, dev), w.to(dev), act(cl(dy), prec, dev)
    assert relerr(ncdhw(Fn.conv3(xd, cin, wd, dims, prec).cpu()), yr) < TOL[prec]
    assert relerr(ncdhw(Fn.conv3(dyd, cout, wd, dims, prec, mode=1).cpu()), xr.grad) < TOL[prec]
    assert relerr(Fn.conv3_wgrad(xd, cin, dyd, cout, dims, cin, cout, prec), wr.grad) < TOL[prec]
    f1 = Fn.conv3_fused(xd, cin, wd, None, dims, prec)          # + InstanceNorm sums flushed across the batch boundary of a long walk
    if f1 is not None:
        st = Fn.instnorm_stats(f1[0], cout, B, D * H * W, cout)
        assert relerr(f1[1].cpu(), st.cpu()) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(432, 768, 3072), (432, 768, 2304), (1000, 768, 3072), (64, 128, 256)])
def test_gemm_ln_bwd_fused_equals_two_steps(pkg, dev, M, N, K):
    """unetr_gemm_bf16_ln_bwd (LayerNorm backward summing the split-K slabs of the GEMM that produced its dy) against the two
    separate entry points: bit-identical dx, bf16 twin and gamma / beta gradients, whether the GEMM is split (batch-2 shapes)
    or not."""
    Fn = pkg.functional
    A = g(M, K, seed=1).to(dev).bfloat16()
    Wt = (g(K, N, seed=2) * 0.05).to(dev).bfloat16()                 # read as the [K, N] operand
    x, gam, bet = g(M, N, seed=3).to(dev), (1 + 0.1 * g(N, seed=4)).to(dev), (0.1 * g(N, seed=5)).to(dev)
    dres = g(M, N, seed=6).to(dev)
    y, mean, rstd = Fn.layernorm_fwd(x, gam, bet)
    dy = torch.empty(M, N, device=dev)
    Fn.gemm_bf16(A, Wt, M, N, K, b_kn=True, C=dy)
    dxb_ref = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    dx_ref, dw_ref, db_ref = Fn.layernorm_bwd(dy, x, gam, mean, rstd, dres=dres, dx_bf16=dxb_ref)
    dxb = torch.empty_like(dxb_ref)
    dx, dw, db = Fn.gemm_ln_bwd_params(A, Wt, M, N, K, x, gam, bet, mean, rstd, dres=dres, dx_bf16=dxb)
    assert torch.equal(dx, dx_ref) and torch.equal(dxb, dxb_ref)
    assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(432, 768, 3072), (1000, 768, 3072), (64, 128, 256)])
def test_gemm_ln_fwd_fused_equals_three_steps(pkg, dev, M, N, K):
    """unetr_gemm_bf16_ln_fwd (the next block's LayerNorm formed by the kernel that sums this GEMM's split-K slabs, adds bias
    and residual and writes the residual stream) against GEMM (+ reduce) followed by LayerNorm: bit-identical C, normalised
    rows, mean and rstd, split or not."""
    Fn = pkg.functional
    A = g(M, K, seed=1).to(dev).bfloat16()
    W = (g(N, K, seed=2) * 0.05).to(dev).bfloat16()
    bias, res = g(N, seed=3).to(dev), g(M, N, seed=4).to(dev)
    gam, bet = (1 + 0.1 * g(N, seed=5)).to(dev), (0.1 * g(N, seed=6)).to(dev)
    c_ref = torch.empty(M, N, device=dev)
    Fn.gemm_bf16(A, W, M, N, K, C=c_ref, bias=bias, res=res, ldr=N)
    yb_ref = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    _, mean_ref, rstd_ref = Fn.layernorm_fwd(c_ref, gam, bet, bf16_out=yb_ref, want_fp32=False)
    c, yb = torch.empty_like(c_ref), torch.empty_like(yb_ref)
    mean, rstd = Fn.gemm_bf16_ln_fwd(A, W, M, N, K, c, gam, bet, yb, bias=bias, res=res, ldr=N)
    assert torch.equal(c, c_ref) and torch.equal(yb, yb_ref)
    assert torch.equal(mean, mean_ref) and torch.equal(rstd, rstd_ref)


# ----------------------------------------------------------------------------- round 4: InstanceNorm work folded into its neighbours
@pytest.mark.gpu
@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("B,dims3,cin,cout,with3", [(2, (8, 8, 16), 16, 16, True), (1, (9, 7, 19), 1, 16, True), (2, (6, 10, 20), 32, 16, True),
                                                     (1, (12, 12, 12), 64, 32, True), (2, (4, 4, 16), 256, 128, True), (2, (20, 12, 40), 16, 16, False),
                                                     (3, (16, 16, 32), 32, 64, True)])
def test_instnorm_finalize_in_apply_prologue(pkg, dev, prec, B, dims3, cin, cout, with3):
    """The statistics finalize folded into the apply kernel (unetr_conv3_fwd_parts + unetr_instnorm_apply_fin) == the separate
    finalize launch (unetr_conv3_fwd_fused + unetr_instnorm_apply): same conv outputs bit for bit, statistics and the applied
    tensor equal up to the summation order of the partial rows."""
    Fn = pkg.functional
    D, H, W = dims3
    V = D * H * W
    image = cin < 8
    x = g(B, D, H, W, cin, seed=1) if image else rq(g(B, D, H, W, cin, seed=1), prec)
    xd = x.to(dev) if image else act(x, prec, dev)
    w, w3 = g(cout, cin, 3, 3, 3, seed=2, scale=0.2).to(dev), (g(cout, cin, 1, 1, 1, seed=3, scale=0.5).to(dev) if with3 else None)
    c, st, c3, st3 = Fn.conv3_fused(xd, cin, w, w3, (B, D, H, W), prec)
    r = Fn.conv3_parts(xd, cin, w, w3, (B, D, H, W), prec)
    assert r is not None
    c_p, part, c3_p, part3, rows = r
    assert torch.equal(c, c_p) and (not with3 or torch.equal(c3, c3_p))
    assert 0 < rows <= Fn.CONV3_MAX_ROWS
    ref1 = Fn.instnorm_apply(c, st, B, V, cout, True)
    got = Fn.instnorm_apply_fin(c_p, part, rows, B, V, cout, True)
    assert got is not None
    y, sa, _ = got
    assert relerr(sa[..., 0], st[..., 0]) < 1e-5 + (st[..., 0].abs().max().item() < 1e-3) and relerr(sa[..., 1], st[..., 1]) < 1e-5
    assert relerr(y, ref1) < (1e-5 if prec == 0 else 8e-3)
    if with3:
        ref2 = Fn.instnorm_apply(c, st, B, V, cout, True, x2=c3, sb=st3)
        y2, sa2, sb2 = Fn.instnorm_apply_fin(c_p, part, rows, B, V, cout, True, x2=c3_p, part_b=part3, rows_b=rows)
        assert relerr(sb2[..., 1], st3[..., 1]) < 1e-5 and torch.equal(sa2, sa)
        assert relerr(y2, ref2) < (1e-5 if prec == 0 else 8e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("B,dims3,C,cout,max_wg", [(2, (8, 8, 16), 16, 16, 0), (1, (9, 7, 19), 16, 16, 0), (2, (6, 10, 20), 32, 32, 0), (1, (12, 12, 12), 64, 64, 0),
                                                   (2, (4, 4, 16), 128, 128, 0), (3, (16, 16, 32), 16, 16, 8), (2, (20, 12, 40), 32, 32, 8)])
def test_conv3_dgrad_with_backward_statistics(pkg, dev, monkeypatch, prec, B, dims3, C, cout, max_wg):
    """Data-gradient conv with the InstanceNorm backward sums in its epilogue + the apply kernel that finalizes them in its
    prologue == plain data-gradient conv + unetr_instnorm_bwd (reduction pass, finalize launch, apply pass), and both == autograd
    of lrelu(instance_norm(c1)) fed through conv2.  max_wg > 0: few persistent workgroups, several batch items per workgroup
    (the partial rows are flushed at batch boundaries)."""
    Fn = pkg.functional
    if max_wg:
        monkeypatch.setenv("UNETR_TEST_MAX_WG", str(max_wg))
    D, H, W = dims3
    V = D * H * W
    dims = (B, D, H, W)
    c1 = rq(g(B, C, D, H, W, seed=1) * 1.3 + 0.2, prec)
    w2 = g(cout, C, 3, 3, 3, seed=2, scale=0.2)
    dc2 = rq(g(B, cout, D, H, W, seed=3), prec)
    c1r = c1.clone().requires_grad_(True)
    a1 = F.leaky_relu(F.instance_norm(c1r, eps=1e-5), 0.01)
    F.conv3d(a1, w2, padding=1).backward(dc2)
    c1d, dc2d, w2d = act(cl(c1), prec, dev), act(cl(dc2), prec, dev), w2.to(dev)
    s1 = Fn.instnorm_stats(c1d, C, B, V, C)
    # unfused
    da1 = Fn.conv3(dc2d, cout, w2d, dims, prec, mode=1)
    dc1_u, _ = Fn.instnorm_bwd(da1, C, c1d, s1, B, V, C, True)
    # fused
    f = Fn.conv3_dgrad_stats(dc2d, w2d, c1d, s1, dims, prec)
    assert f is not None
    da1_f, part, rows = f
    assert torch.equal(da1_f, da1)
    dc1_f = Fn.instnorm_bwd_apply_fin(da1_f, C, c1d, s1, part, rows, 2, B, V, C, True)
    assert dc1_f is not None
    tol = 3e-5 if prec == 0 else 2e-2
    assert relerr(dc1_f, dc1_u) < (2e-5 if prec == 0 else 1e-2)          # (bf16: the fused sums see the fp32 accumulators, not their bf16 image)
    assert relerr(ncdhw(dc1_f.float().cpu()), c1r.grad) < tol
    assert relerr(ncdhw(dc1_u.float().cpu()), c1r.grad) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("B,S,C", [(2, 12, 16), (1, 20, 32), (2, 6, 128), (2, 24, 64)])
def test_instnorm_bwd_folded_finalize(pkg, dev, monkeypatch, prec, B, S, C):
    """unetr_instnorm_bwd with the finalize of its partial sums in the apply kernel's prologue (default) == with the separate
    finalize launch (UNETR_IN_FIN=0), single and dual form"""
    Fn = pkg.functional
    V = S ** 3
    x, x2, dy = rq(g(B, C, S, S, S, seed=1) * 1.5 + 0.3, prec), rq(g(B, C, S, S, S, seed=2) * 0.7 - 0.2, prec), rq(g(B, C, S, S, S, seed=3), prec)
    xd, x2d, dyd = act(cl(x), prec, dev), act(cl(x2), prec, dev), act(cl(dy), prec, dev)
    sa, sb = Fn.instnorm_stats(xd, C, B, V, C), Fn.instnorm_stats(x2d, C, B, V, C)
    dx, dx2 = Fn.instnorm_bwd(dyd, C, xd, sa, B, V, C, True, x2=x2d, sb=sb)
    dx1, _ = Fn.instnorm_bwd(dyd, C, xd, sa, B, V, C, True)
    monkeypatch.setenv("UNETR_IN_FIN", "0")
    dxu, dx2u = Fn.instnorm_bwd(dyd, C, xd, sa, B, V, C, True, x2=x2d, sb=sb)
    dx1u, _ = Fn.instnorm_bwd(dyd, C, xd, sa, B, V, C, True)
    t = 1e-5 if prec == 0 else 8e-3
    assert relerr(dx, dxu) < t and relerr(dx2, dx2u) < t and relerr(dx1, dx1u) < t
    xr, x2r = x.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    F.leaky_relu(F.instance_norm(xr, eps=1e-5) + F.instance_norm(x2r, eps=1e-5), 0.01).backward(dy)
    t2 = 2e-5 if prec == 0 else 5e-3
    assert relerr(ncdhw(dx.float().cpu()), xr.grad) < t2 and relerr(ncdhw(dx2.float().cpu()), x2r.grad) < t2
    xr1 = x.clone().requires_grad_(True)
    F.leaky_relu(F.instance_norm(xr1, eps=1e-5), 0.01).backward(dy)
    assert relerr(ncdhw(dx1.float().cpu()), xr1.grad) < t2


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("B,dims3,cin,cout", [(2, (8, 8, 16), 32, 16), (1, (12, 12, 12), 64, 32), (2, (6, 6, 6), 256, 128), (2, (16, 16, 16), 1, 16),
                                              (1, (8, 12, 16), 4, 16), (3, (9, 7, 19), 1, 16), (2, (20, 12, 40), 1, 32)])
def test_resblock_in_fusion_levels(pkg, dev, monkeypatch, prec, B, dims3, cin, cout):
    """MONAI UnetResBlock forward + backward with the InstanceNorm work folded into its neighbours (UNETR_AMD_IN_FUSE=3)
    against the round-3 launch sequence (=0): outputs and all four gradients agree to rounding.  The 1- and 4-channel cases are
    the block on the image: at level 3 its 1x1x1 branch is never stored (formed from the image inside the block-end kernels, its
    weight gradient summed inside the backward apply)."""
    Fn = pkg.functional
    D, H, W = dims3
    image = cin < 8
    x0 = g(B, D, H, W, cin, seed=1) if image else rq(g(B, D, H, W, cin, seed=1), prec)
    w = [g(cout, cin, 3, 3, 3, seed=2, scale=0.2), g(cout, cout, 3, 3, 3, seed=3, scale=0.2), g(cout, cin, 1, 1, 1, seed=4, scale=0.5)]
    dout = act(rq(g(B, D, H, W, cout, seed=5), prec), prec, dev)
    res = {}
    for level in ("0", "3"):
        monkeypatch.setenv("UNETR_AMD_IN_FUSE", level)
        xd = (x0.to(dev) if image else act(x0, prec, dev)).requires_grad_(not image)
        ws = [t.to(dev).requires_grad_(True) for t in w]
        out = Fn.ResBlockFn.apply(xd, ws[0], ws[1], ws[2], prec)
        out.backward(dout)
        res[level] = [out.detach().float()] + [t.grad.float() for t in ws] + ([] if image else [xd.grad.float()])
    tol = 2e-4 if prec == 0 else 3e-2
    for a, b in zip(res["0"], res["3"]):
        assert relerr(a, b) < tol
