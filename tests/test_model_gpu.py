"""End-to-end parity of the HIP UNETR (through the reference's nn.Module interface) against the CPU oracle:
same state_dict, same seeded synthetic volume -> logits, enc4, Dice/CE terms and parameter gradients.
north_star tolerance: 1e-3 relative fp32 on logits and Dice (fp32 mode).  bf16 mode is reported with its own
looser bound (bf16 operands, fp32 accumulate, 12 residual blocks)."""
import os

import pytest
import torch

from util import relerr

pytestmark = pytest.mark.gpu

C1 = dict(in_channels=1, out_channels=2, img_size=(32, 32, 32), feature_size=16, hidden_size=128, mlp_dim=512,
          num_heads=4, pos_embed="perceptron", norm_name="instance", res_block=True)
GRAD_KEYS = ["vit.patch_embedding.patch_embeddings.1.weight", "vit.patch_embedding.position_embeddings",
             "vit.blocks.0.attn.qkv.weight", "vit.blocks.5.norm1.weight", "vit.blocks.11.mlp.linear2.weight",
             "vit.blocks.11.mlp.linear1.bias", "vit.norm.bias", "encoder1.layer.conv1.conv.weight",
             "encoder1.layer.conv3.conv.weight", "encoder2.blocks.1.conv.weight", "encoder4.transp_conv_init.conv.weight",
             "decoder5.transp_conv.conv.weight", "decoder5.conv_block.conv1.conv.weight",
             "decoder2.conv_block.conv2.conv.weight", "decoder2.conv_block.conv3.conv.weight", "out.conv.conv.weight",
             "out.conv.conv.bias"]


def _pair(pkg, dev, cfg, seed=0, ref_dtype=torch.float64):
    """Oracle in fp64 by default: the deep-layer gradients of this network are ill-conditioned (InstanceNorm
    over 4^3 voxels at the bottleneck) -- the fp32 oracle itself sits ~5e-4 from the fp64 one, so fp64 is the
    fair yardstick for two fp32 implementations with different summation orders."""
    from oracle.unetr_oracle import OracleUNETR
    torch.manual_seed(seed)
    ref = OracleUNETR(**cfg)
    hip = pkg.UNETR(**cfg)
    hip.load_state_dict(ref.state_dict(), strict=True)
    return ref.to(ref_dtype), hip.to(dev)


def cosine(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return (a @ b / (a.norm() * b.norm()).clamp_min(1e-300)).item()


def _run(pkg, dev, cfg, batch, precision, freeze=False):
    from oracle.unetr_oracle import oracle_dice_ce_terms, synthetic_volume
    ref, hip = _pair(pkg, dev, cfg)
    hip.precision = precision
    x, y = synthetic_volume(batch, cfg["in_channels"], cfg["img_size"][0], cfg["out_channels"], seed=7)
    rd = next(ref.parameters()).dtype
    enc4_r, logits_r = ref(x.to(rd), freeze_encoder=freeze)
    d_r, c_r = oracle_dice_ce_terms(logits_r, y.to(rd))
    (d_r + c_r).backward()
    enc4, logits = hip(x.to(dev), freeze_encoder=freeze)
    terms = pkg.DiceCELoss(to_onehot_y=True, softmax=True).terms(logits, y.to(dev))
    terms[0].backward()
    torch.cuda.synchronize()
    gr = dict(ref.named_parameters())
    gh = dict(hip.named_parameters())
    return dict(enc4=(enc4, enc4_r), logits=(logits, logits_r), dice=(terms[1], d_r), ce=(terms[2], c_r)), gr, gh


def test_c1_fp32_parity(pkg, dev):
    outs, gr, gh = _run(pkg, dev, C1, 2, "fp32")
    for k, (a, b) in outs.items():
        assert relerr(a, b) < 1e-3, k
    assert gh["vit.patch_embedding.cls_token"].grad is None
    errs = {k: relerr(gh[k].grad, gr[k].grad) for k in GRAD_KEYS}
    print(errs)
    for k, e in errs.items():
        assert e < 5e-3, (k, e)
    # every parameter with an oracle gradient has one here too
    for k, p in gr.items():
        assert (p.grad is None) == (gh[k].grad is None), k


def test_c1_bf16x3_parity(pkg, dev):
    """the second tolerance-grade mode (bf16x3: fp32 storage, operands split into bf16 hi/lo pairs, fp32 accumulation) at config[0]
    against the fp64 oracle: 1e-3 on logits / enc4 / Dice / CE (measured 2e-5 / 7e-6 / 2e-7 / 5e-7).  Gradients: this geometry
    normalises over 2^3 voxels at its bottleneck, which amplifies the ~2^-17 product error of the split by ~1e3 on everything
    upstream of it (measured 1-2e-2 element-wise where the fp32 mode has ~1e-3), so they are held by direction and a looser
    element bound; the full-size geometry (12^3 voxels at the bottleneck) holds cosine 0.9999 (test_c2_full_size_fp32_parity)."""
    outs, gr, gh = _run(pkg, dev, C1, 2, "bf16x3")
    errs = {k: relerr(a, b) for k, (a, b) in outs.items()}
    print(errs)
    for k, e in errs.items():
        assert e < 1e-3, (k, e)
    gerr = {k: (relerr(gh[k].grad, gr[k].grad), cosine(gh[k].grad, gr[k].grad)) for k in GRAD_KEYS}
    print(gerr)
    for k, (e, c) in gerr.items():
        assert e < 5e-2 and c > 0.9995, (k, e, c)


def test_c1_bf16_bounded(pkg, dev):
    outs, gr, gh = _run(pkg, dev, C1, 1, "bf16")
    for k, (a, b) in outs.items():
        assert relerr(a, b) < 5e-2, k
    cos = {k: cosine(gh[k].grad, gr[k].grad) for k in GRAD_KEYS}
    print(cos)
    for k, c in cos.items():
        assert c > 0.97, (k, c)


@pytest.mark.parametrize("flat", [False, True])
def test_bf16_token_rows_not_multiple_of_8(pkg, dev, flat):
    """bf16 mode at 48^3 / batch 1: 27 token rows.  The forward GEMMs have no row-count requirement, but the bf16 grouped weight
    gradient needs rows % 8 == 0 -- such a shape must take the fp32 weight-gradient route end to end (the patch embedding once
    kept no fp32 patch matrix for it and raised in backward), with and without the flat arenas."""
    cfg = dict(C1, img_size=(48, 48, 48))
    from oracle.unetr_oracle import oracle_dice_ce_terms, synthetic_volume
    ref, hip = _pair(pkg, dev, cfg)
    hip.precision = "bf16"
    if flat:
        hip.use_flat_buffers()
    x, y = synthetic_volume(1, 1, 48, 2, seed=11)
    _, logits_r = ref(x.double())
    d_r, c_r = oracle_dice_ce_terms(logits_r, y.double())
    (d_r + c_r).backward()
    _, logits = hip(x.to(dev))
    pkg.DiceCELoss(to_onehot_y=True, softmax=True)(logits, y.to(dev)).backward()
    torch.cuda.synchronize()
    assert relerr(logits, logits_r) < 5e-2
    gr, gh = dict(ref.named_parameters()), dict(hip.named_parameters())
    for k in ("vit.patch_embedding.patch_embeddings.1.weight", "vit.patch_embedding.position_embeddings", "vit.blocks.0.attn.qkv.weight",
              "vit.blocks.11.mlp.linear2.weight", "decoder2.conv_block.conv2.conv.weight"):
        assert gh[k].grad is not None and cosine(gh[k].grad, gr[k].grad) > 0.97, k
    if flat:
        pkg.functional.clear_grad_sinks()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_c1_fourteen_classes(pkg, dev, precision):
    """The reference's DEFAULT head -- n_classes = 14 (BTCV, unetr_segmentation_3d.py:303) -- forward + DiceCE + backward in both
    precision modes: the out conv's weight gradient for more than 4 classes takes the generic kernel, on bf16-stored feature
    maps in bf16 mode."""
    outs, gr, gh = _run(pkg, dev, dict(C1, out_channels=14), 1, precision)
    tol = 1e-3 if precision == "fp32" else 5e-2
    for k, (a, b) in outs.items():
        assert relerr(a, b) < tol, k
    assert gh["out.conv.conv.weight"].grad.shape == (14, 16, 1, 1, 1)
    for k in ("out.conv.conv.weight", "out.conv.conv.bias", "decoder2.conv_block.conv2.conv.weight", "vit.blocks.11.mlp.linear2.weight"):
        if precision == "fp32":
            assert relerr(gh[k].grad, gr[k].grad) < 5e-3, k
        else:
            assert cosine(gh[k].grad, gr[k].grad) > 0.97, k


def test_freeze_encoder(pkg, dev):
    outs, gr, gh = _run(pkg, dev, C1, 1, "fp32", freeze=True)
    for k, (a, b) in outs.items():
        assert relerr(a, b) < 1e-3, k
    for k, p in gr.items():
        assert (p.grad is None) == (gh[k].grad is None), k
        if p.grad is not None:
            # batch 1: InstanceNorm over 4^3 voxels at the bottleneck is even worse conditioned than in the batch-2 test
            assert relerr(gh[k].grad, p.grad) < 1e-2, k
    assert gh["vit.blocks.0.attn.qkv.weight"].grad is None and gh["decoder2.transp_conv.conv.weight"].grad is not None


def test_against_committed_golden(pkg, dev):
    """HIP path vs tests/golden/c1_seed0.npz (fp64-oracle vectors, see make_golden.py) -- nothing from oracle/ runs."""
    import os
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_seed0.npz"))
    torch.manual_seed(0)
    hip = pkg.UNETR(**C1)   # MONAI-equivalent seeded initialisation order is shared with the oracle's constructor
    assert abs(sum(p.double().sum().item() for p in hip.parameters()) - float(g["weight_checksum"])) < 1e-6
    hip = hip.to(dev)
    hip.precision = "fp32"
    x = torch.from_numpy(g["x"]).to(dev)
    y = torch.from_numpy(g["y"].astype(np.float32)).to(dev)
    enc4, logits = hip(x)
    t = pkg.DiceCELoss(to_onehot_y=True, softmax=True).terms(logits, y)
    t[0].backward()
    ls = logits.detach()[0, :, ::4, ::4, ::4].cpu().numpy()
    es = enc4.detach()[0, ::16, ::2, ::2, ::2].cpu().numpy()
    assert np.abs(ls - g["logits_sub"]).max() < 1e-3 * float(g["logits_absmax"])
    assert np.abs(es - g["enc4_sub"]).max() < 1e-3 * float(g["enc4_absmax"])
    assert abs(t[1].item() - float(g["dice"])) < 1e-3 * float(g["dice"])
    assert abs(t[2].item() - float(g["ce"])) < 1e-3 * float(g["ce"])
    gh = dict(hip.named_parameters())
    for k in g.files:
        if k.startswith("grad:"):
            ref = g[k]
            got = gh[k[5:]].grad.flatten()[::7].cpu().numpy()
            assert np.abs(got - ref).max() <= 1e-2 * np.abs(ref).max(), k


def test_logits_only_and_train_step(pkg, dev):
    """monai.networks.nets.UNETR call convention + two AdamW steps reduce the loss and track the oracle."""
    from oracle.unetr_oracle import OracleUNETR, oracle_train_step, synthetic_volume
    torch.manual_seed(1)
    ref = OracleUNETR(**C1)
    hip = pkg.UNETRLogits(**C1)
    hip.load_state_dict(ref.state_dict(), strict=True)
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    hip = hip.to(dev)
    x, y = synthetic_volume(1, 1, 32, 2, seed=3)
    o_ref = torch.optim.AdamW(ref.parameters(), lr=1e-4, weight_decay=1e-5)
    o_hip = torch.optim.AdamW(hip.parameters(), lr=1e-4, weight_decay=1e-5)
    crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)
    xd, yd = x.to(dev), y.to(dev)
    for _ in range(2):
        l_ref = oracle_train_step(ref, o_ref, x, y)
        logit_map = hip(xd)
        loss = crit(logit_map, yd)
        loss.backward()
        o_hip.step()
        o_hip.zero_grad()
        assert relerr(loss, l_ref) < 1e-3
    # the two AdamW steps moved the weights the same way.  (Compared as update DIRECTION: Adam's first steps are ~ lr * sign(g),
    # so a gradient element near zero may flip its whole 1e-4 step on a last-bit difference -- an element-wise bound on the
    # weights themselves, relative to max |w| ~ 3e-2, sits right at that noise level.)
    k = "decoder2.conv_block.conv1.conv.weight"
    w0 = sd0[k]
    du_r, du_h = ref.state_dict()[k] - w0, hip.state_dict()[k].cpu() - w0
    assert cosine(du_h, du_r) > 0.99 and relerr(hip.state_dict()[k], ref.state_dict()[k]) < 1e-2


def test_flat_buffers_match_per_tensor_path(pkg, dev):
    """use_flat_buffers(): gradients land in the arena (param.grad is an arena view; Linear weight grads come from the
    grouped end-of-backward launch), the unused cls_token keeps grad None, and flat AdamW == torch.optim.AdamW when fed
    the same gradients (Adam amplifies rounding differences, so it is checked step by step on identical inputs)."""
    from oracle.unetr_oracle import synthetic_volume
    torch.manual_seed(3)
    a = pkg.UNETRLogits(**C1).to(dev)
    b = pkg.UNETRLogits(**C1).to(dev)
    b.load_state_dict(a.state_dict())
    flat = b.use_flat_buffers()
    crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)
    x, y = synthetic_volume(1, 1, 32, 2, seed=5)
    x, y = x.to(dev), y.to(dev)
    la = crit(a(x), y); la.backward()
    lb = crit(b(x), y); lb.backward()
    assert torch.equal(la, lb)
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    lo = flat["grad"].data_ptr()
    for k, p in pa.items():
        if p.grad is None:
            assert pb[k].grad is None, k
        else:
            assert relerr(pb[k].grad, p.grad) < 2e-5, k       # grouped wgrad: other tile shape, no split-K
            assert lo <= pb[k].grad.data_ptr() < lo + flat["grad"].numel() * 4, k
    # optimiser: torch reference on clones, fed b's own gradients every step
    ref = {k: p.detach().clone().requires_grad_(True) for k, p in pb.items()}
    o_ref = torch.optim.AdamW(list(ref.values()), lr=1e-3, weight_decay=1e-2)
    o_b = pkg.AdamW(b.parameters(), lr=1e-3, weight_decay=1e-2, flat=flat)
    for it in range(3):
        if it:
            crit(b(x), y).backward()
        for k, p in pb.items():
            ref[k].grad = None if p.grad is None else p.grad.detach().clone()
        o_ref.step()
        o_b.step()
        o_b.zero_grad()
        for k, p in pb.items():
            assert relerr(p, ref[k]) < 1e-5, (it, k)
    assert torch.equal(pb["vit.patch_embedding.cls_token"], torch.zeros_like(pb["vit.patch_embedding.cls_token"]))
    pkg.functional.clear_grad_sinks()


def test_bf16x3_flat_arena_weight_gradients(pkg, dev):
    """bf16x3 mode with the flat arenas: the ViT weight gradients run on the bf16 grouped kernel over (hi, lo) row stacks
    (functional._launch_deferred) -- same gradients as the per-tensor path's generic split-operand GEMM, to rounding"""
    from oracle.unetr_oracle import synthetic_volume
    torch.manual_seed(3)
    a = pkg.UNETRLogits(**C1).to(dev)
    b = pkg.UNETRLogits(**C1).to(dev)
    b.load_state_dict(a.state_dict())
    a.precision = b.precision = "bf16x3"
    flat = b.use_flat_buffers()
    crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)
    x, y = synthetic_volume(2, 1, 32, 2, seed=5)
    x, y = x.to(dev), y.to(dev)
    la = crit(a(x), y); la.backward()
    lb = crit(b(x), y); lb.backward()
    assert torch.equal(la, lb)
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    for k, p in pa.items():
        if p.grad is None:
            assert pb[k].grad is None, k
        else:
            assert relerr(pb[k].grad, p.grad) < 1e-4, k
    pkg.functional.clear_grad_sinks(flat["state"])


def test_bf16x3_weight_word_shadow(pkg, dev, monkeypatch):
    """bf16x3 mode with flat arenas: the Linear GEMMs read the weights from the word shadow (functional.weight_x3: [hi | lo << 16]
    words next to the parameter arena, re-derived by one launch after every AdamW step).  Forward / backward with the shadow equal the
    in-register split of the fp32 weights (UNETR_AMD_X3_WORDS=0) to rounding; the shadow follows an optimizer step, an in-place torch
    write (version counter) and load_state_dict."""
    from oracle.unetr_oracle import synthetic_volume
    Fn = pkg.functional
    torch.manual_seed(7)
    m = pkg.UNETRLogits(**C1).to(dev)
    m.precision = "bf16x3"
    flat = m.use_flat_buffers()
    opt = pkg.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-2, flat=flat)
    crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)
    x, y = synthetic_volume(2, 1, 32, 2, seed=8)
    x, y = x.to(dev), y.to(dev)

    def fwd_bwd(words):
        monkeypatch.setenv("UNETR_AMD_X3_WORDS", words)
        opt.zero_grad(set_to_none=True)
        out = m(x)
        crit(out, y).backward()
        return out.detach().clone(), flat["grad"].clone()

    def words_match():
        w = m.vit.blocks[0].mlp.linear1.weight
        sh = Fn.weight_x3(w)
        assert sh is not None and sh.dtype == torch.int32 and sh.shape == w.shape
        hi = (sh << 16).view(torch.float32)
        lo = (sh & -65536).view(torch.float32)
        assert relerr(hi + lo, w) < 2e-5

    for phase in range(4):
        o1, g1 = fwd_bwd("1")
        assert flat.get("shadow_x3") is not None
        words_match()
        o0, g0 = fwd_bwd("0")
        assert relerr(o1, o0) < 1e-5 and relerr(g1, g0) < 1e-4, phase
        if phase == 0:
            opt.step()                                         # AdamW kernels write the arena: the optimizer re-derives the words
        elif phase == 1:
            with torch.no_grad():
                m.vit.blocks[0].mlp.linear1.weight.mul_(1.25)  # torch writes a parameter: version counter
        elif phase == 2:
            sd = {k: v * 0.9 for k, v in m.state_dict().items()}
            m.load_state_dict(sd)
    Fn.clear_grad_sinks(flat["state"])


@pytest.mark.parametrize("comm_dtype", [torch.float32, torch.bfloat16])
def test_data_parallel_arena_update(pkg, dev, comm_dtype):
    """The N>1 update of bench.py on one device: AdamW.step_reduced reads the 'all-reduced' gradient SUM of a simulated
    2-rank job (both ranks hold the same gradient) from the communication buffer in <= 3 planned pieces and averages on
    the fly.  It must match AdamW.step() fed the (communication-dtype rounded) gradients bit for bit, keep the unused
    cls_token untouched, and keep the bf16 weight shadow in step with the fp32 master."""
    from oracle.unetr_oracle import synthetic_volume
    Fn = pkg.functional
    torch.manual_seed(4)
    crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)
    x, y = synthetic_volume(1, 1, 32, 2, seed=6)
    x, y = x.to(dev), y.to(dev)
    results = []
    for mode in ("plain", "reduced"):
        torch.manual_seed(4)
        m = pkg.UNETRLogits(**C1).to(dev)
        flat = m.use_flat_buffers()
        opt = pkg.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-2, flat=flat)
        for it in range(2):
            crit(m(x), y).backward()
            g = flat["grad"]
            if mode == "plain":
                g.copy_(g.to(comm_dtype).float())               # what the other path sees after the cast
                opt.step()
            else:
                buf = (g.to(comm_dtype) * 2)                    # sum over two identical ranks (exact in both dtypes)
                plan = opt.plan_reduced(max_elems=(flat["total"] + 2) // 3)
                assert len(plan["runs"]) >= 3
                seen = []
                opt.step_reduced(plan, buf, 0.5, before_run=lambda k, lo, hi: seen.append((k, lo, hi)))
                assert [k for k, _, _ in seen] == list(range(len(plan["runs"])))
            opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        results.append((flat["param"].clone(), flat["shadow"].clone()))
        assert torch.equal(flat["shadow"].float(), flat["param"].bfloat16().float())
        Fn.clear_grad_sinks()
    assert torch.equal(results[0][0], results[1][0])
    assert torch.equal(results[0][1], results[1][1])


C2 = dict(in_channels=1, out_channels=4, img_size=(96, 96, 96), feature_size=16, hidden_size=768, mlp_dim=3072,
          num_heads=12, pos_embed="perceptron", norm_name="instance", res_block=True)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_c2_full_size_fp32_parity(pkg, dev, precision):
    """BASELINE config[1] geometry (96^3, hidden 768, 12 heads, 4 classes) at batch 1 against the fp32 CPU oracle: logits / enc4 /
    Dice / CE within north_star's 1e-3; gradients by cosine (the fp32 oracle is itself noisy).  Both tolerance-grade modes: fp32
    (exact fp32 MFMA chains) and bf16x3 (fp32 storage, split bf16 operands: ~16-bit products at four times the fp32 matrix rate)."""
    from oracle.unetr_oracle import oracle_dice_ce_terms, synthetic_volume
    ref, hip = _pair(pkg, dev, C2, ref_dtype=torch.float32)
    hip.precision = precision
    x, y = synthetic_volume(1, 1, 96, 4, seed=11)
    enc4_r, logits_r = ref(x)
    d_r, c_r = oracle_dice_ce_terms(logits_r, y)
    (d_r + c_r).backward()
    enc4, logits = hip(x.to(dev))
    t = pkg.DiceCELoss(to_onehot_y=True, softmax=True).terms(logits, y.to(dev))
    t[0].backward()
    assert relerr(logits, logits_r) < 1e-3 and relerr(enc4, enc4_r) < 1e-3
    assert relerr(t[1], d_r) < 1e-3 and relerr(t[2], c_r) < 1e-3
    gr, gh = dict(ref.named_parameters()), dict(hip.named_parameters())
    for k in ["vit.blocks.0.attn.qkv.weight", "vit.blocks.11.mlp.linear1.weight", "encoder1.layer.conv2.conv.weight",
              "decoder5.conv_block.conv1.conv.weight", "decoder2.conv_block.conv1.conv.weight", "decoder2.transp_conv.conv.weight",
              "out.conv.conv.weight"]:
        assert cosine(gh[k].grad, gr[k].grad) > 0.9999, k


def test_c4_160_forward_parity(pkg, dev):
    """BASELINE config[3] geometry: 160^3 input, 1000 tokens (attention tiles with a masked tail, 10^3 token grid)."""
    from oracle.unetr_oracle import synthetic_volume
    cfg = dict(C2, img_size=(160, 160, 160), hidden_size=192, mlp_dim=384, num_heads=3, out_channels=3)
    ref, hip = _pair(pkg, dev, cfg, ref_dtype=torch.float32)
    hip.precision = "fp32"
    x, _ = synthetic_volume(1, 1, 160, 3, seed=5)
    with torch.no_grad():
        enc4_r, logits_r = ref(x)
        enc4, logits = hip(x.to(dev))
    assert logits.shape == (1, 3, 160, 160, 160) and enc4.shape == (1, 128, 20, 20, 20)
    assert relerr(logits, logits_r) < 1e-3 and relerr(enc4, enc4_r) < 1e-3


def test_task01_multilabel_4channel(pkg, dev):
    """SURVEY 8(f) rank 3 -- the 4-channel MR task (unetr_segmentation_3d.py:310-312, 477-482): in_channels = 4 and
    DiceCELoss(to_onehot_y=False, sigmoid=True) on a multi-label target.  Train-step parity on a small geometry against
    the fp64 oracle, then the 128^3 geometry (patch_dim 16 384, 512 tokens) forward against the fp32 oracle."""
    from oracle.unetr_oracle import oracle_dice_ce_terms, synthetic_volume
    cfg = dict(C1, in_channels=4, out_channels=4)
    ref, hip = _pair(pkg, dev, cfg, seed=3)
    hip.precision = "fp32"
    x, _ = synthetic_volume(2, 4, 32, 4, seed=11)
    target = (torch.rand(2, 4, 32, 32, 32, generator=torch.Generator().manual_seed(12)) < 0.3).float()
    rd = next(ref.parameters()).dtype
    _, logits_r = ref(x.to(rd))
    d_r, c_r = oracle_dice_ce_terms(logits_r, target.to(rd), to_onehot_y=False, softmax=False, sigmoid=True)
    (d_r + c_r).backward()
    _, logits = hip(x.to(dev))
    t = pkg.DiceCELoss(to_onehot_y=False, sigmoid=True).terms(logits, target.to(dev))
    t[0].backward()
    assert relerr(logits, logits_r) < 1e-3 and relerr(t[1], d_r) < 1e-3 and relerr(t[2], c_r) < 1e-3
    gr, gh = dict(ref.named_parameters()), dict(hip.named_parameters())
    for k, p in gr.items():
        if p.grad is None:
            assert gh[k].grad is None, k
        else:
            assert relerr(gh[k].grad, p.grad) < 1e-2, k
    del ref, hip
    big = dict(C2, in_channels=4, out_channels=4, img_size=(128, 128, 128), hidden_size=192, mlp_dim=384, num_heads=3)
    ref, hip = _pair(pkg, dev, big, ref_dtype=torch.float32)
    hip.precision = "fp32"
    x, _ = synthetic_volume(1, 4, 128, 4, seed=13)
    with torch.no_grad():
        enc4_r, logits_r = ref(x)
        enc4, logits = hip(x.to(dev))
    assert logits.shape == (1, 4, 128, 128, 128) and enc4.shape == (1, 128, 16, 16, 16)
    assert relerr(logits, logits_r) < 1e-3 and relerr(enc4, enc4_r) < 1e-3


def test_c5_pretraining_steps(pkg, dev):
    """BASELINE config[4]: the two stages of unetr_ranking_pretraining_3d.py:238-296 on a [4, ...] batch -- 'feat'
    (loss on enc4, everything trains) and 'recon' (loss on the logits with freeze_encoder=True) -- vs the oracle."""
    from oracle.unetr_oracle import oracle_bt_loss, oracle_extract_triplets, synthetic_volume
    ref, hip = _pair(pkg, dev, C1, seed=2)
    hip.precision = "fp32"
    x, _ = synthetic_volume(4, 1, 32, 2, seed=9)
    rd = next(ref.parameters()).dtype
    for stage, dim, T in (("feat", 2, 0.1), ("recon", 4, 0.1)):
        ref.zero_grad()
        hip.zero_grad()
        if stage == "feat":
            inp_r, _ = ref(x.to(rd))
            inp_h, _ = hip(x.to(dev))
            init_idx = 0                      # enc4 is 4^3 here: partition size 1
        else:
            _, inp_r = ref(x.to(rd), freeze_encoder=True)
            _, inp_h = hip(x.to(dev), freeze_encoder=True)
            init_idx = 3                      # logits are 32^3: partition size 8
        f1, f2 = torch.split(inp_r, [2, 2], dim=0)
        l_r = oracle_bt_loss(*oracle_extract_triplets(f1, f2, dim, init_idx), T)
        l_r.backward()
        l_h = pkg.ranking_loss(inp_h, dim, init_idx, T, kind="ranking")
        l_h.backward()
        assert relerr(l_h, l_r) < 1e-3, stage
        gr, gh = dict(ref.named_parameters()), dict(hip.named_parameters())
        for k, p in gr.items():
            assert (p.grad is None) == (gh[k].grad is None), (stage, k)
        keys = ["decoder2.conv_block.conv1.conv.weight", "out.conv.conv.weight"] if stage == "recon" else \
            ["encoder4.transp_conv_init.conv.weight", "vit.blocks.9.mlp.linear1.weight", "vit.patch_embedding.patch_embeddings.1.weight"]
        for k in keys:
            assert cosine(gh[k].grad, gr[k].grad) > 0.999, (stage, k)


def test_sliding_window_inference_and_dice_metric(pkg, dev):
    """SURVEY 8(f) rank 2 -- the validation step (unetr_segmentation_3d.py:103-132): sliding-window inference with the
    reference's arguments (roi = crop^3, sw_batch_size 4, default overlap 0.25, and the 0.8 of :694-695) over volumes larger
    than, equal to and smaller than the window, then argmax/one-hot post-processing and DiceMetric "mean" / "mean_batch" -- against the CPU
    oracle running the oracle model."""
    from oracle.unetr_oracle import (oracle_dice_metric, oracle_post_label, oracle_post_pred,
                                     oracle_sliding_window_inference, synthetic_volume)
    ref, hip = _pair(pkg, dev, C1, seed=4, ref_dtype=torch.float32)
    hip.precision = "fp32"
    ref_pred = lambda w: ref(w)[1]
    for size, overlap in ((48, 0.25), (40, 0.5), (48, 0.8), (32, 0.25), (24, 0.25)):      # 0.8: the plots' call, unetr_segmentation_3d.py:694-695
        x, y = synthetic_volume(2, 1, size, 2, seed=20 + size)
        with torch.no_grad():
            out_r = oracle_sliding_window_inference(x, (32, 32, 32), 4, ref_pred, overlap=overlap)
        out_h = pkg.sliding_window_inference(x.to(dev), (32, 32, 32), 4, hip, overlap=overlap)
        assert out_h.shape == out_r.shape == (2, 2, size, size, size)
        assert relerr(out_h, out_r) < 1e-3, size
        # metric on the oracle's own predictions (so argmax ties / near-ties cannot differ), both call conventions
        pr, lr = oracle_post_pred(out_r, 2), oracle_post_label(y, 2)
        for red in ("mean", "mean_batch"):
            raw_r, val_r = oracle_dice_metric(pr, lr, red)
            m = pkg.DiceMetric(include_background=True, reduction=red, get_not_nans=False)
            raw_h = m(y_pred=[t for t in pr.to(dev)], y=[t for t in lr.to(dev)])
            assert torch.allclose(raw_h.cpu(), raw_r, atol=1e-6, equal_nan=True)
            assert torch.allclose(m.aggregate().cpu(), val_r, atol=1e-6)
            m.reset()
            m(out_r.to(dev), y.to(dev), from_logits=True)           # fused argmax + one-hot
            assert torch.allclose(m.aggregate().cpu(), val_r, atol=1e-6)
        # north_star: Dice of the HIP prediction within 1e-3 of the reference's
        m = pkg.DiceMetric()
        m(out_h, y.to(dev), from_logits=True)
        assert abs(m.aggregate().item() - oracle_dice_metric(pr, lr)[1].item()) < 1e-3
    # an absent class gives NaN for that item/class and drops out of the averages
    y0 = torch.zeros(2, 1, 24, 24, 24)
    pr, lr = oracle_post_pred(out_r, 2), oracle_post_label(y0, 2)
    raw_r, val_r = oracle_dice_metric(pr, lr, "mean")
    m = pkg.DiceMetric()
    raw_h = m(pr.to(dev), lr.to(dev))
    assert torch.isnan(raw_h[:, 1]).all() and torch.allclose(raw_h.cpu(), raw_r, atol=1e-6, equal_nan=True)
    assert torch.allclose(m.aggregate().cpu(), val_r, atol=1e-6)


def test_training_trajectory_tracks_oracle(pkg, dev):
    """24 optimiser steps of the reference loop (unetr_segmentation_3d.py:220-226) on a fixed batch: the HIP path (flat
    arenas + this package's AdamW, i.e. exactly what bench.py runs) must follow the CPU oracle's loss curve.  Adam
    amplifies rounding differences step after step (the fp32 CPU oracle itself drifts 1e-3 .. 4e-3 from the fp64 one over
    40 steps, depending on the width of the model), so the yardstick is the fp64 oracle: tight over the first 10 steps,
    bounded over all 24."""
    from oracle.unetr_oracle import OracleUNETR, oracle_train_step, synthetic_volume
    Fn = pkg.functional
    x, y = synthetic_volume(2, 1, 32, 2, seed=31)
    torch.manual_seed(5)
    ref = OracleUNETR(**C1)
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    ref = ref.double()
    o_ref = torch.optim.AdamW(ref.parameters(), lr=3e-4, weight_decay=1e-5)
    c64 = [float(oracle_train_step(ref, o_ref, x.double(), y.double())) for _ in range(24)]
    assert c64[-1] < 0.95 * c64[0]                         # the problem actually trains
    crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)
    xd, yd = x.to(dev), y.to(dev)
    for precision, tight, loose in (("fp32", 2e-3, 2e-2), ("bf16", 1e-2, 3e-2)):
        hip = pkg.UNETRLogits(**C1)
        hip.load_state_dict(sd0, strict=True)
        hip = hip.to(dev)
        hip.precision = precision
        flat = hip.use_flat_buffers()
        opt = pkg.AdamW(hip.parameters(), lr=3e-4, weight_decay=1e-5, flat=flat)
        devs = []
        for ref_loss in c64:
            loss = crit(hip(xd), yd)
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            devs.append(abs(float(loss.detach()) - ref_loss) / abs(ref_loss))
        assert max(devs[:10]) < tight and max(devs) < loose, (precision, max(devs[:10]), max(devs))
        Fn.clear_grad_sinks()


def test_derived_weight_copies_follow_every_update(pkg, dev):
    """bf16 weight shadows and packed conv weights are caches of the fp32 masters: whichever way the masters change --
    this package's AdamW (which maintains the copies itself), a torch optimizer, in-place edits, load_state_dict, a
    re-pointed .data -- the next forward must see the new values (checked against the oracle carrying the same weights)."""
    from oracle.unetr_oracle import OracleUNETR, synthetic_volume
    torch.manual_seed(7)
    ref = OracleUNETR(**C1)
    hip = pkg.UNETR(**C1).to(dev)
    hip.precision = "bf16"                      # both caches are in play in bf16 mode
    hip.load_state_dict(ref.state_dict(), strict=True)
    x, y = synthetic_volume(1, 1, 32, 2, seed=8)
    xd, yd = x.to(dev), y.to(dev)
    crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)

    def check(tag):
        ref.load_state_dict({k: v.detach().cpu() for k, v in hip.state_dict().items()}, strict=True)
        with torch.no_grad():
            lr = ref(x)[1]
            lh = hip(xd)[1]
        assert relerr(lh, lr) < 5e-2, (tag, relerr(lh, lr))
        return lh

    base = check("initial")
    # 1. this package's AdamW, per-tensor mode (shadows / packs registered by the forward above become optimizer-maintained)
    opt = pkg.AdamW(hip.parameters(), lr=5e-2, weight_decay=0.0)
    crit(hip(xd)[1], yd).backward()
    opt.step(); opt.zero_grad(set_to_none=True)
    after = check("AdamW")
    assert relerr(after, base) > 1e-2                      # the weights really moved
    # 2. a torch optimizer (bumps the version counters)
    sgd = torch.optim.SGD(hip.parameters(), lr=0.5)
    crit(hip(xd)[1], yd).backward()
    sgd.step(); sgd.zero_grad(set_to_none=True)
    check("SGD")
    # 3. in-place edits of one Linear and one conv weight
    with torch.no_grad():
        hip.vit.blocks[0].mlp.linear1.weight.mul_(1.5)
        hip.decoder2.conv_block.conv1.conv.weight.add_(0.05)
        hip.encoder1.layer.conv3.conv.weight.mul_(-1.0)
    check("in-place")
    # 4. load_state_dict of fresh weights
    torch.manual_seed(9)
    hip.load_state_dict(OracleUNETR(**C1).state_dict(), strict=True)
    check("load_state_dict")
    # 5. re-pointed .data (same shape, same device, version counter unchanged)
    w = hip.decoder3.conv_block.conv2.conv.weight
    w.data = (w.data * 2.0).clone()
    q = hip.vit.blocks[1].attn.qkv.weight
    q.data = (q.data * 0.5).clone()
    check("re-pointed .data")
    # 6. in-place writes THROUGH .data (neither version counter nor address changes).  The copies are not
    #    optimizer-maintained at this point (torch touched every parameter since AdamW last stepped), so each forward
    #    pass re-derives them (functional.begin_forward): seen without any call
    hip.decoder4.conv_block.conv1.conv.weight.data.mul_(1.7)
    hip.vit.blocks[2].mlp.linear2.weight.data.mul_(0.3)
    check(".data in-place, copies not optimizer-maintained")
    # 7. the same kind of write once this package's AdamW maintains the copies: invalidate_weight_shadows() is mandatory
    crit(hip(xd)[1], yd).backward()
    opt.step(); opt.zero_grad(set_to_none=True)
    check("AdamW again")
    hip.decoder4.conv_block.conv1.conv.weight.data.mul_(0.5)
    hip.vit.blocks[2].mlp.linear2.weight.data.mul_(2.0)
    pkg.invalidate_weight_shadows()
    check(".data in-place + invalidate_weight_shadows()")


# bounds of the benched configuration = 2x what was measured on MI355X (gpurun_out/r3_c2_parity.log; printed by the test with -s):
# a regression that doubles any of these errors fails.  north_star's 1e-3 is met in fp32 mode (test_c2_full_size_fp32_parity),
# not in this -- the benched -- bf16 mode.
# measured (round 3): logits 1.91e-2, enc4 5.1e-3, Dice term 2.0e-5, CE term 8.7e-5, lowest gradient cosine 0.99365
# (vit.patch_embedding.position_embeddings; next 0.9972), loss of optimizer steps 1 / 3 / 4 within 6e-5 / 9e-5 / 2e-5 of the oracle's.
C2_BOUNDS = dict(logits=4e-2, enc4=1.1e-2, dice=1e-4, ce=2e-4, cosine=0.99, loss0=2e-4, loss_traj=2e-4)
C2_COSINE_ALLOW = {}       # tensor name -> its own lower bound (none needed: the measured floor over all 164 tensors is above 0.99)


def test_c2_bench_path_bf16_parity(pkg, dev):
    """BASELINE config[1] EXACTLY as bench.py runs it -- bf16 mode, batch 2, flat arenas, this package's AdamW, the whole
    step replayed as a captured hipGraph (train_step.TrainStep) -- against the fp32 CPU oracle with the same weights
    and the same synthetic volumes.  bf16 bounds (bf16 operands, fp32 accumulate, 12 residual blocks + 5 conv stages):
    logits / enc4 / Dice / CE terms within C2_BOUNDS (= 2x the measured errors), EVERY parameter gradient (164 tensors) by
    cosine >= 0.99 (no exceptions needed), and the loss of optimizer steps 1, 3, 4 within 2e-4 of the oracle's trajectory."""
    from oracle.unetr_oracle import OracleUNETR, oracle_dice_ce_terms, oracle_train_step, synthetic_volume
    torch.manual_seed(1234)
    ref = OracleUNETR(**C2)
    hip = pkg.UNETRLogits(**C2)
    hip.load_state_dict(ref.state_dict(), strict=True)
    hip = hip.to(dev)
    hip.precision = "bf16"
    x, y = synthetic_volume(2, 1, 96, 4, seed=1234)
    xd, yd = x.to(dev), y.to(dev)
    # forward quantities and all gradients at the initial weights (eager, same kernels the graph replays)
    flat = hip.use_flat_buffers()
    crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)
    enc4, logits = pkg.UNETR.forward(hip, xd)
    terms = crit.terms(logits, yd)
    terms[0].backward()
    torch.cuda.synchronize()
    enc4_r, logits_r = ref(x)
    d_r, c_r = oracle_dice_ce_terms(logits_r, y)
    (d_r + c_r).backward()
    meas = dict(logits=relerr(logits, logits_r), enc4=relerr(enc4, enc4_r), dice=relerr(terms[1], d_r), ce=relerr(terms[2], c_r))
    print("measured forward errors:", meas)
    assert meas["logits"] < C2_BOUNDS["logits"] and meas["enc4"] < C2_BOUNDS["enc4"], meas
    assert meas["dice"] < C2_BOUNDS["dice"] and meas["ce"] < C2_BOUNDS["ce"], meas
    gr, gh = dict(ref.named_parameters()), dict(hip.named_parameters())
    n, worst = 0, {}
    for k, p in gr.items():
        if p.grad is None:
            assert gh[k].grad is None, k
            continue
        c = cosine(gh[k].grad, p.grad)
        worst[k] = c
        n += 1
    assert n == 164
    print("lowest cosines:", sorted(worst.items(), key=lambda kv: kv[1])[:8])
    low = {k: c for k, c in worst.items() if c < C2_BOUNDS["cosine"] and k not in C2_COSINE_ALLOW}
    assert not low, low
    assert all(worst[k] > v for k, v in C2_COSINE_ALLOW.items()), {k: worst[k] for k in C2_COSINE_ALLOW}
    ref.zero_grad()
    hip.zero_grad(set_to_none=True)
    # the benched step: AdamW on arenas, graph replay; losses of steps 1..4 against the oracle's
    o_ref = torch.optim.AdamW(ref.parameters(), lr=1e-4, weight_decay=1e-5)
    opt = pkg.AdamW(hip.parameters(), lr=1e-4, weight_decay=1e-5, flat=flat)
    # (fuse_update as bench.py runs it at N=1: AdamW of the ViT Linear weights in the weight-gradient epilogue; held bit for bit to
    # the separate optimizer launch by test_staged_backward_equals_single_pass)
    step = pkg.TrainStep(hip, crit, opt, xd, yd, use_graph=True, warmup=2, fuse_update=True)     # 2 eager steps, then capture (1 more: capture runs nothing)
    assert step.graphs is not None and len(step.graphs) == 1 and step.fuse
    l_ref = [float(oracle_train_step(ref, o_ref, x, y)) for _ in range(4)]
    traj = [abs(float(step.first_loss) - l_ref[0]) / l_ref[0]]
    step.run()                                                                     # optimizer step 3 (graph replay)
    traj.append(abs(float(step.loss.detach()) - l_ref[2]) / l_ref[2])
    step.run()
    traj.append(abs(float(step.loss.detach()) - l_ref[3]) / l_ref[3])
    print("measured loss-trajectory errors (steps 1, 3, 4):", traj)
    assert traj[0] < C2_BOUNDS["loss0"] and max(traj[1:]) < C2_BOUNDS["loss_traj"], (traj, l_ref)
    assert l_ref[3] < l_ref[0]
    flat["state"].clear()


@pytest.mark.parametrize("size", ["c1", "c2"])
def test_staged_backward_equals_single_pass(pkg, dev, size):
    """The data-parallel launch form (forward_staged + 5 backward passes, per-pass reduce slots, AdamW per reduced piece,
    4 hipGraphs, hand-over through the host or by stream wait) must give the SAME parameters as the single-graph step: same kernels, same order of every floating-point
    sum (the only fan-out sums have two terms).  Run on one rank (the all-reduce of a 1-rank job is the identity).
    size c2 = BASELINE configs[2]'s per-rank workload exactly (96^3, hidden 768, 4 classes, bf16, batch 2, fp32 gradient
    communication): the launch form the 8-GPU bench runs, held bit for bit to the single-GPU step of configs[1]."""
    from oracle.unetr_oracle import synthetic_volume
    cfg, S, ncls, lr = (C2, 96, 4, 1e-4) if size == "c2" else (C1, 32, 2, 1e-3)
    x, y = synthetic_volume(2, 1, S, ncls, seed=41)
    xd, yd = x.to(dev), y.to(dev)
    res = {}
    modes = (("single", "staged_eager", "staged_graph", "staged_graph_streamwait", "staged_bf16comm", "staged_bf16comm_nofuse", "staged_bf16comm_eager", "overlap_one_graph",
              "fused_eager", "fused_graph") if size == "c1"
             else ("single", "staged_graph", "fused_graph"))
    for mode in modes:
        torch.manual_seed(11)
        m = pkg.UNETRLogits(**cfg).to(dev)
        m.precision = "bf16"
        flat = m.use_flat_buffers()
        opt = pkg.AdamW(m.parameters(), lr=lr, weight_decay=1e-5, flat=flat)
        crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)
        step = pkg.TrainStep(m, crit, opt, xd, yd, use_graph=not mode.endswith("_eager"), data_parallel=mode.startswith("staged"),
                             comm_dtype=torch.bfloat16 if "bf16comm" in mode else torch.float32, warmup=2,
                             overlap_update=mode == "overlap_one_graph", fuse_update=mode.startswith("fused"), fuse_comm="nofuse" not in mode,
                             handover="stream" if mode.endswith("streamwait") else "host")
        if "bf16comm" in mode:               # bf16(dW) of the ViT weights straight from the weight-gradient epilogue into the comm buffer
            assert step.fuse_comm == ("nofuse" not in mode) and (step._comm_fuse is not None) == step.fuse_comm
        if mode.startswith("fused"):         # AdamW of the 48 Linear weights + patch embedding rides on the weight-gradient launch
            assert step.fuse and step._fuse_pattern is not None and "epilogue" in step.launch or mode == "fused_eager"
        if mode == "overlap_one_graph":      # the same passes + side-stream AdamW as ONE graph (the side stream is a branch of it)
            assert step.one_graph and len(step.graphs) == 1
        if mode.startswith("staged"):
            assert len(step.pieces) == 5 and len(step.pieces[4]) == 2
            if size == "c2":        # the tail that cannot overlap with backward: block 0 + patch embedding, 41 MB of 370 MB
                tail = sum(hi - lo for lo, hi in step.pieces[4]) * 4
                assert tail < 45e6 and sum(hi - lo for st in step.pieces for lo, hi in st) * 4 > 365e6
            if not mode.endswith("_eager"):      # pass 0 hands nothing over: it shares a graph with pass 1
                assert len(step.graphs) == 4 and step.graph_passes == [[0, 1], [2], [3], [4]]
        for _ in range(3):
            step.run()
        torch.cuda.synchronize()
        res[mode] = (flat["param"].clone(), float(step.loss.detach()), opt._flat_state[0].clone(), opt._flat_state[1].clone(), flat["shadow"].clone())
        flat["state"].clear()
        del step, opt, m, flat
    if "staged_eager" in res:
        assert torch.equal(res["single"][0], res["staged_eager"][0])
        assert torch.equal(res["single"][0], res["overlap_one_graph"][0])
    assert torch.equal(res["single"][0], res["staged_graph"][0])
    assert res["single"][1] == res["staged_graph"][1]
    if "staged_graph_streamwait" in res:                        # hand-over by cross-stream wait instead of through the host
        assert torch.equal(res["single"][0], res["staged_graph_streamwait"][0])
    for mode in (k for k in res if k.startswith("fused")):      # the fused optimizer epilogue: masters, both moments, bf16 shadows, loss
        for k in (0, 2, 3, 4):
            assert torch.equal(res["single"][k], res[mode][k]), (mode, k)
        assert res["single"][1] == res[mode][1]
    if "staged_bf16comm" in res:
        assert relerr(res["staged_bf16comm"][0], res["single"][0]) < 1e-2       # bf16-rounded gradients: close, not equal
        for other in ("staged_bf16comm_nofuse", "staged_bf16comm_eager"):        # the epilogue's bf16 gradients = the cast pass's
            for k in (0, 2, 3, 4):
                assert torch.equal(res["staged_bf16comm"][k], res[other][k]), (other, k)


@pytest.mark.parametrize("mode,size", [("eager", "c1"), ("graph", "c1"), ("graph", "c2")])
def test_two_rank_data_parallel_step(pkg, dev, tmp_path, mode, size):
    """The data-parallel TrainStep with a REAL 2-rank all-reduce, on one GPU: two processes (gloo carries the CUDA gradient
    pieces -- RCCL refuses two ranks on one device), each with its shard of a batch of 4, against one process stepping on the
    whole batch.  Both ranks must end with bit-identical parameters, and the averaged-gradient update must be the update of
    the batch-4 step (DiceCE is a mean over batch items, InstanceNorm is per item: equal up to summation order).
    size c2 = two ranks of BASELINE configs[2] at its own workload (96^3, hidden 768, bf16, batch 2 per rank, 5 captured passes)."""
    import socket
    import subprocess
    import sys
    from oracle.unetr_oracle import synthetic_volume
    cfg, S, ncls, lr = (C2, 96, 4, 1e-4) if size == "c2" else (C1, 32, 2, 1e-3)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dp_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(port), str(tmp_path), mode, size], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=900)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-2000:] for o in outs)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(2))
    assert torch.equal(r0["param"], r1["param"])
    # the same job in one process: batch 4, single-graph step (warm-up steps count as steps: compare equal step counts)
    x, y = synthetic_volume(4, 1, S, ncls, seed=77)
    torch.manual_seed(11)
    m = pkg.UNETRLogits(**cfg).to(dev)
    m.precision = "bf16"
    flat = m.use_flat_buffers()
    p0 = flat["param"].clone()
    opt = pkg.AdamW(m.parameters(), lr=lr, weight_decay=1e-5, flat=flat)
    step = pkg.TrainStep(m, pkg.DiceCELoss(to_onehot_y=True, softmax=True), opt, x.to(dev), y.to(dev), use_graph=False, warmup=1)
    while step.eager_steps < r0["steps"] + (2 if mode == "graph" else 0):
        step.run()
    torch.cuda.synchronize()
    upd_dp, upd_1 = r0["param"].to(dev) - p0, flat["param"] - p0
    cos = torch.nn.functional.cosine_similarity(upd_dp.double(), upd_1.double(), dim=0)
    # (Adam's first steps are ~ lr * sign(g): at 92 M parameters many gradients sit near zero, where bf16 rounding of a different
    # batch split flips signs -- the full-size bound is looser than the 32^3 one)
    assert cos > (0.98 if size == "c1" else 0.90), float(cos)
    assert relerr(r0["param"].to(dev), flat["param"]) < 5e-3
    flat["state"].clear()


@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_two_rank_ranking_pretraining_step(pkg, dev, tmp_path, mode):
    """BASELINE configs[4] in its data-parallel form with a REAL 2-rank all-reduce (two processes on one GPU, gloo): per rank a
    "feat" TrainStep (loss on enc4: staged backward, per-pass all-reduce under the passes that follow) and a "recon" TrainStep
    (encoder frozen: one backward pass, its gradient runs are the pieces), two rounds.  Both ranks end bit-identical, the frozen
    step's pieces lie in the decoder's arena range, and the update is the update of ONE process that averages the gradients of
    the two ranks' batches itself (per-tensor gradients accumulated over the two [4, ...] batches, each loss halved)."""
    import socket
    import subprocess
    import sys
    from oracle.unetr_oracle import synthetic_volume
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dp_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(port), str(tmp_path), mode, "c5"], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=900)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(2))
    assert torch.equal(r0["param"], r1["param"])
    # one process, joint batch, the same number of feat / recon updates
    x, _ = synthetic_volume(8, 1, 32, 2, seed=77)
    torch.manual_seed(11)
    m = pkg.UNETR(**C1).to(dev)
    m.precision = "bf16"
    flat = m.use_flat_buffers()                                       # (only for the arena geometry and the flat parameter vector)
    names = [n for n, _ in m.named_parameters()]
    dec_lo = min(o for n, o in zip(names, flat["offsets"]) if n.startswith("decoder") or n.startswith("out."))
    for lo, hi in r0["pieces"][1][0]:
        assert dec_lo <= lo < hi <= flat["total"]                    # the frozen pass communicates decoder / out gradients only
    assert len(r0["pieces"][0]) == 5                                  # feat: the five staged passes
    p0 = flat["param"].clone()
    flat["state"].clear()                                             # per-tensor gradients from here on: they accumulate over two batches
    opt = pkg.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5)
    xr = [torch.cat([x[2 * r:2 * r + 2], x[4 + 2 * r:4 + 2 * r + 2]]).to(dev) for r in range(2)]      # the two ranks' batches
    nf, nr = r0["steps"][0] + (2 if mode == "graph" else 0), r0["steps"][1] + (2 if mode == "graph" else 0)
    assert nf == nr
    for _ in range(nf):
        for xb in xr:
            enc4, _ = m(xb)
            (0.5 * pkg.ranking_loss(enc4, 2, 0, 0.1, kind="ranking")).backward()
        opt.step(); opt.zero_grad(set_to_none=True)
        for xb in xr:
            _, logits = m(xb, freeze_encoder=True)
            (0.5 * pkg.ranking_loss(logits, 4, 3, 0.1, kind="ranking")).backward()
        opt.step(); opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    upd_dp, upd_1 = r0["param"].to(dev) - p0, flat["param"] - p0
    cos = torch.nn.functional.cosine_similarity(upd_dp.double(), upd_1.double(), dim=0)
    assert cos > 0.95, float(cos)
    assert relerr(r0["param"].to(dev), flat["param"]) < 1e-2


def test_next_block_layernorm_ride_is_exact(pkg, dev, monkeypatch):
    """UNETR_AMD_LN_RIDE=1 (norm1 of block i+1 formed by block i's last split-K reduction): logits, loss and every gradient
    bit-identical to the default launch form."""
    from oracle.unetr_oracle import synthetic_volume
    x, y = synthetic_volume(2, 1, 96, 4, seed=5)
    xd, yd = x.to(dev), y.to(dev)
    res = []
    for ride in ("0", "1"):
        monkeypatch.setenv("UNETR_AMD_LN_RIDE", ride)
        torch.manual_seed(3)
        m = pkg.UNETRLogits(**C2).to(dev)
        m.precision = "bf16"
        logits = m(xd)
        loss = pkg.DiceCELoss(to_onehot_y=True, softmax=True)(logits, yd)
        loss.backward()
        res.append([logits.detach().clone(), loss.detach().clone()] + [p.grad.clone() for p in m.parameters() if p.grad is not None])
        del m
    assert len(res[0]) == len(res[1])
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_two_models_and_failed_backward(pkg, dev):
    """Re-entrancy of the arena fast path: two models with flat arenas in one process keep separate deferred
    weight-gradient queues (interleaved forward / backward of A and B give each the gradients it gets alone), and a
    backward pass that raises half-way does not leave stale work armed: the next step's gradients are right."""
    from oracle.unetr_oracle import synthetic_volume
    x, y = synthetic_volume(1, 1, 32, 2, seed=43)
    xd, yd = x.to(dev), y.to(dev)
    crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)

    def make(seed):
        torch.manual_seed(seed)
        m = pkg.UNETRLogits(**C1).to(dev)
        m.precision = "bf16"
        return m, m.use_flat_buffers()

    a, fa = make(1)
    crit(a(xd), yd).backward()
    ga_alone = fa["grad"].clone()
    a.zero_grad(set_to_none=True)
    b, fb = make(2)
    crit(b(xd), yd).backward()
    gb_alone = fb["grad"].clone()
    b.zero_grad(set_to_none=True)
    assert fa["state"] is not fb["state"]
    la = crit(a(xd), yd)                 # interleaved: both forwards first, then B's backward, then A's
    lb = crit(b(xd), yd)
    lb.backward()
    la.backward()
    torch.cuda.synchronize()
    assert torch.equal(fa["grad"], ga_alone) and torch.equal(fb["grad"], gb_alone)
    a.zero_grad(set_to_none=True)
    b.zero_grad(set_to_none=True)

    # a backward pass that dies half-way: pass 1 of the staged form runs vit.norm, block 11 and block 10 (which queue their
    # weight gradients for the end-of-pass grouped launch), then a hook on hidden state 9 raises -- the autograd engine
    # drops its end-of-pass callbacks, so the queue stays armed with stale work
    _, logits, stages = a.forward_staged(xd)
    crit(logits, yd).backward()
    hs9 = stages[0][0][0]

    def boom(g):
        raise RuntimeError("boom")
    hs9.register_hook(boom)
    with pytest.raises(RuntimeError, match="boom"):
        torch.autograd.backward([r for r, _ in stages[0]], [l.grad for _, l in stages[0]])
    st = fa["state"]
    assert st.defer["armed"] and len(st.defer["wgrad_b"]) + len(st.defer["wgrad"]) >= 8       # blocks 11 and 10 were queued
    del stages, logits, hs9
    a.zero_grad(set_to_none=True)
    fa["grad"].zero_()
    crit(a(xd), yd).backward()           # a clean step afterwards
    torch.cuda.synchronize()
    assert torch.equal(fa["grad"], ga_alone)
    fa["state"].clear()
    fb["state"].clear()


def test_encoder_checkpointing_is_exact(pkg, dev):
    """UNETR.encoder_checkpointing (BASELINE config[3]: activation checkpointing on the encoder): the transformer blocks keep
    only their input and recompute their forward inside backward -- same kernels on the same inputs, so loss and every gradient
    are bit-identical to the run that keeps the activations, in both precision modes."""
    from oracle.unetr_oracle import synthetic_volume
    x, y = synthetic_volume(2, 1, 32, 2, seed=51)
    xd, yd = x.to(dev), y.to(dev)
    crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)
    for precision in ("fp32", "bf16"):
        grads = []
        for ckpt in (False, True):
            torch.manual_seed(13)
            m = pkg.UNETRLogits(**C1).to(dev)
            m.precision = precision
            m.encoder_checkpointing = ckpt
            loss = crit(m(xd), yd)
            loss.backward()
            torch.cuda.synchronize()
            grads.append((loss.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
        assert torch.equal(grads[0][0], grads[1][0])
        assert grads[0][1].keys() == grads[1][1].keys()
        for k in grads[0][1]:
            assert torch.equal(grads[0][1][k], grads[1][1][k]), (precision, k)


def test_c4_160_checkpointed_train_step_parity(pkg, dev):
    """BASELINE config[3] AT ITS WORKLOAD: 160^3 input, hidden 768 / 12 heads / mlp 3072 (1000 tokens: attention chunk loop),
    encoder activation checkpointing, batch 1 -- forward + DiceCE + backward against the fp32 CPU oracle.  fp32 mode: logits /
    enc4 / Dice / CE within north_star's 1e-3, gradients by cosine; bf16 mode (the benched one): bf16 bounds."""
    from oracle.unetr_oracle import OracleUNETR, oracle_dice_ce_terms, synthetic_volume
    cfg = dict(C2, img_size=(160, 160, 160))
    torch.manual_seed(21)
    ref = OracleUNETR(**cfg)
    x, y = synthetic_volume(1, 1, 160, 4, seed=23)
    enc4_r, logits_r = ref(x)
    d_r, c_r = oracle_dice_ce_terms(logits_r, y)
    (d_r + c_r).backward()
    gr = {k: p.grad for k, p in ref.named_parameters()}
    keys = ["vit.blocks.0.attn.qkv.weight", "vit.blocks.6.mlp.linear1.weight", "vit.blocks.11.attn.out_proj.weight", "vit.norm.weight",
            "vit.patch_embedding.patch_embeddings.1.weight", "encoder1.layer.conv2.conv.weight", "encoder3.blocks.0.conv.weight",
            "decoder5.conv_block.conv1.conv.weight", "decoder2.conv_block.conv1.conv.weight", "decoder2.transp_conv.conv.weight",
            "out.conv.conv.weight"]
    crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)
    for precision, tol, cmin in (("fp32", 1e-3, 0.9999), ("bf16", 5e-2, 0.98)):
        hip = pkg.UNETR(**cfg)
        hip.load_state_dict(ref.state_dict(), strict=True)
        hip = hip.to(dev)
        hip.precision = precision
        hip.encoder_checkpointing = True
        enc4, logits = hip(x.to(dev))
        t = crit.terms(logits, y.to(dev))
        t[0].backward()
        torch.cuda.synchronize()
        assert logits.shape == (1, 4, 160, 160, 160) and enc4.shape == (1, 128, 20, 20, 20)
        assert relerr(logits, logits_r) < tol and relerr(enc4, enc4_r) < tol, precision
        assert relerr(t[1], d_r) < max(tol / 5, 1e-3) and relerr(t[2], c_r) < max(tol / 5, 1e-3), precision
        gh = dict(hip.named_parameters())
        for k in keys:
            assert cosine(gh[k].grad, gr[k]) > cmin, (precision, k, cosine(gh[k].grad, gr[k]))
        for k, g in gr.items():
            assert (g is None) == (gh[k].grad is None), k
        del hip, enc4, logits, t


def test_c5_ranking_pretraining_at_96(pkg, dev):
    """BASELINE config[4] AT ITS WORKLOAD (unetr_ranking_pretraining_3d.py:238-296): a [4, 1, 96^3] batch (2 volumes x 2
    transforms), for each of the three slice axes a 'feat' pass (Bradley-Terry loss on enc4 [4,128,12^3], everything trains) and a
    'recon' pass (loss on the logits [4,2,96^3] = 16 slices of 18 432 features, encoder frozen) against the fp32 CPU oracle:
    loss values for all six passes, gradients (cosine) for the two passes of slice axis 3, None-pattern of the frozen pass."""
    from oracle.unetr_oracle import OracleUNETR, oracle_bt_loss, oracle_extract_triplets, synthetic_volume
    cfg = dict(C2, out_channels=2)
    torch.manual_seed(31)
    ref = OracleUNETR(**cfg)
    hip = pkg.UNETR(**cfg)
    hip.load_state_dict(ref.state_dict(), strict=True)
    hip = hip.to(dev)
    hip.precision = "fp32"
    x, _ = synthetic_volume(4, 1, 96, 2, seed=33)
    xd = x.to(dev)
    for axis in (2, 3, 4):
        for stage, init_idx in (("feat", 1), ("recon", 7)):
            ref.zero_grad()
            hip.zero_grad()
            grads = axis == 3
            with torch.set_grad_enabled(grads):
                if stage == "feat":
                    inp_r, _ = ref(x)
                    inp_h, _ = hip(xd)
                else:
                    _, inp_r = ref(x, freeze_encoder=True)
                    _, inp_h = hip(xd, freeze_encoder=True)
                f1, f2 = torch.split(inp_r, [2, 2], dim=0)
                l_r = oracle_bt_loss(*oracle_extract_triplets(f1, f2, axis, init_idx), 0.1)
                l_h = pkg.ranking_loss(inp_h, axis, init_idx, 0.1, kind="ranking")
            assert relerr(l_h, l_r) < 1e-3, (axis, stage, float(l_h), float(l_r))
            if not grads:
                continue
            l_r.backward()
            l_h.backward()
            gr, gh = dict(ref.named_parameters()), dict(hip.named_parameters())
            for k, p in gr.items():
                assert (p.grad is None) == (gh[k].grad is None), (stage, k)
            keys = ["decoder2.conv_block.conv1.conv.weight", "decoder5.transp_conv.conv.weight", "out.conv.conv.weight"] if stage == "recon" else \
                ["encoder4.transp_conv_init.conv.weight", "vit.blocks.9.mlp.linear1.weight", "vit.blocks.0.attn.qkv.weight",
                 "vit.patch_embedding.patch_embeddings.1.weight"]
            for k in keys:
                assert cosine(gh[k].grad, gr[k].grad) > 0.999, (stage, k, cosine(gh[k].grad, gr[k].grad))
            if stage == "recon":
                assert gh["vit.blocks.0.attn.qkv.weight"].grad is None and gh["encoder1.layer.conv1.conv.weight"].grad is None


def test_capture_mode_vs_foreign_thread_queries(pkg, dev, tmp_path):
    """TrainStep captures its hipGraphs in ``thread_local`` capture mode because a live process group's watchdog thread polls
    events (hipEventQuery) while the step is being captured.  Deterministic form of that situation: a second Python thread
    hammers ``Event.query()`` / ``Stream.query()`` while this thread captures a graph of HIP-extension kernels.  Under
    thread_local mode the capture must complete and replay correctly.  The same under the default ``global`` mode runs in a
    child process (an invalidated capture can poison the context) and is only REPORTED (with -s): round 2 saw one abort in six
    1-rank RCCL runs under global mode and no log of it survives; on MI355X / ROCm 7.2 / torch 2.10 this deterministic form
    COMPLETES under global mode too (round 3), so foreign-thread event queries alone do not explain that abort -- the cause
    stays unknown, thread_local stays as the conservative setting, and this test pins that it works."""
    import subprocess
    import sys
    import threading
    Fn = pkg.functional
    x = torch.randn(432, 768, device=dev)
    gam, bet = torch.ones(768, device=dev), torch.zeros(768, device=dev)
    ev = torch.cuda.Event()
    ev.record()
    stop, errs, polls = threading.Event(), [], [0]

    def poll():
        try:
            while not stop.is_set():
                ev.query()
                torch.cuda.current_stream().query()
                polls[0] += 1
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))
    t = threading.Thread(target=poll, daemon=True)
    t.start()
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            Fn.layernorm_fwd(x, gam, bet)             # warm-up outside capture
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            for _ in range(50):
                y, _, _ = Fn.layernorm_fwd(x, gam, bet)
        g.replay()
        torch.cuda.synchronize()
    finally:
        stop.set()
        t.join(timeout=10)
    assert not errs, errs
    assert polls[0] > 0
    assert relerr(y, torch.nn.functional.layer_norm(x, (768,), gam, bet)) < 1e-5
    # the same under global capture mode, in a child: outcome recorded, not asserted (it is timing dependent)
    child = tmp_path / "global_mode.py"
    child.write_text(f"""
import importlib, sys, threading, torch
sys.path.insert(0, {repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))})
pkg = importlib.import_module("3dmedicalimagesegmentation_amd"); Fn = pkg.functional
dev = torch.device("cuda:0")
x = torch.randn(432, 768, device=dev); gam, bet = torch.ones(768, device=dev), torch.zeros(768, device=dev)
ev = torch.cuda.Event(); ev.record()
stop, errs = threading.Event(), []
def poll():
    try:
        while not stop.is_set():
            ev.query(); torch.cuda.current_stream().query()
    except Exception as e:
        errs.append(repr(e))
t = threading.Thread(target=poll, daemon=True); t.start()
out = "capture completed"
try:
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        Fn.layernorm_fwd(x, gam, bet)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="global"):
        for _ in range(50):
            Fn.layernorm_fwd(x, gam, bet)
    g.replay(); torch.cuda.synchronize()
except Exception as e:
    out = "capture failed: " + repr(e)[:300]
stop.set(); t.join(timeout=10)
print("GLOBAL MODE:", out, "| polling thread errors:", errs[:1])
""")
    r = subprocess.run([sys.executable, str(child)], capture_output=True, text=True, timeout=300)
    print("global capture mode with a polling thread ->", (r.stdout.strip().splitlines() or ["<no output>"])[-1], "| rc", r.returncode)
