"""CPU tests of the oracle (the checker): reference-derived anchors of SURVEY.md 8c, self-consistency, and the
committed golden fixture.  PARITY UNPINNED against MONAI 0.6.0 itself (not importable here)."""
import os

import numpy as np
import pytest
import torch

from oracle.unetr_oracle import (OracleUNETR, oracle_bt_loss, oracle_contrastive_loss, oracle_dice_ce_terms,
                                 oracle_extract_triplets, synthetic_volume)

C1 = dict(in_channels=1, out_channels=2, img_size=(32, 32, 32), feature_size=16, hidden_size=128, mlp_dim=512,
          num_heads=4, pos_embed="perceptron", norm_name="instance", res_block=True)
C2 = dict(in_channels=1, out_channels=4, img_size=(96, 96, 96), feature_size=16, hidden_size=768, mlp_dim=3072,
          num_heads=12, pos_embed="perceptron", norm_name="instance", res_block=True)
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_seed0.npz")


def test_param_counts_and_schema():
    # SURVEY.md 8c anchors (2): 92 452 868 parameters at config 2, 5 209 570 at config 1
    m2 = OracleUNETR(**C2)
    assert sum(p.numel() for p in m2.parameters()) == 92452868
    assert sum(p.numel() for p in OracleUNETR(**C1).parameters()) == 5209570
    sd = m2.state_dict()
    assert len(sd) == 165
    expect = {
        "vit.patch_embedding.position_embeddings": (1, 216, 768), "vit.patch_embedding.cls_token": (1, 1, 768),
        "vit.patch_embedding.patch_embeddings.1.weight": (768, 4096), "vit.blocks.11.attn.qkv.weight": (2304, 768),
        "vit.blocks.0.mlp.linear1.weight": (3072, 768), "vit.norm.bias": (768,),
        "encoder1.layer.conv3.conv.weight": (16, 1, 1, 1, 1), "encoder2.transp_conv_init.conv.weight": (768, 32, 2, 2, 2),
        "encoder2.blocks.1.conv.weight": (32, 32, 2, 2, 2), "encoder3.blocks.0.conv.weight": (64, 64, 2, 2, 2),
        "decoder5.conv_block.conv1.conv.weight": (128, 256, 3, 3, 3), "decoder2.transp_conv.conv.weight": (32, 16, 2, 2, 2),
        "out.conv.conv.weight": (4, 16, 1, 1, 1), "out.conv.conv.bias": (4,),
    }
    for k, shp in expect.items():
        assert tuple(sd[k].shape) == shp, k
    assert "vit.blocks.0.attn.qkv.bias" not in sd


def test_constructor_exceptions():
    # unetr.py:60-67
    with pytest.raises(AssertionError):
        OracleUNETR(**{**C1, "dropout_rate": 1.5})
    with pytest.raises(AssertionError):
        OracleUNETR(**{**C1, "num_heads": 3})
    with pytest.raises(KeyError):
        OracleUNETR(**{**C1, "pos_embed": "fourier"})


def test_shapes_and_freeze():
    torch.manual_seed(0)
    m = OracleUNETR(**C1)
    x, y = synthetic_volume(2, 1, 32, 2, seed=1)
    enc4, logits = m(x, freeze_encoder=True)
    assert enc4.shape == (2, 128, 4, 4, 4) and logits.shape == (2, 2, 32, 32, 32)
    assert not enc4.requires_grad
    d, c = oracle_dice_ce_terms(logits, y)
    (d + c).backward()
    g = dict(m.named_parameters())
    assert g["vit.blocks.0.attn.qkv.weight"].grad is None and g["encoder1.layer.conv1.conv.weight"].grad is None
    assert g["decoder5.transp_conv.conv.weight"].grad is not None and g["out.conv.conv.bias"].grad is not None


def test_perceptron_patch_order_matches_einops():
    einops = pytest.importorskip("einops")
    from oracle.unetr_oracle import _PerceptronPatches
    x = torch.randn(2, 3, 32, 16, 48)
    ref = einops.rearrange(x, "b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)", p1=16, p2=16, p3=16)
    assert torch.equal(_PerceptronPatches((16, 16, 16))(x), ref)


def test_dice_ce_against_torch_primitives():
    torch.manual_seed(0)
    logits = torch.randn(2, 3, 5, 6, 7)
    y = torch.randint(0, 3, (2, 1, 5, 6, 7)).float()
    d, c = oracle_dice_ce_terms(logits, y)
    p = logits.softmax(1)
    oh = torch.nn.functional.one_hot(y[:, 0].long(), 3).permute(0, 4, 1, 2, 3).float()
    dice = 0.0
    for b in range(2):
        for k in range(3):
            i, gsum, ps = (p[b, k] * oh[b, k]).sum(), oh[b, k].sum(), p[b, k].sum()
            dice += 1 - (2 * i + 1e-5) / (gsum + ps + 1e-5)
    assert abs(d.item() - dice.item() / 6) < 1e-6
    assert abs(c.item() - torch.nn.functional.cross_entropy(logits, y[:, 0].long()).item()) < 1e-6


def test_golden_fixture_regression():
    g = np.load(GOLD)
    torch.manual_seed(0)
    m = OracleUNETR(**C1).double()   # the fixture comes from the fp64 oracle (see make_golden.py)
    assert abs(sum(p.sum().item() for p in m.parameters()) - float(g["weight_checksum"])) < 1e-6, \
        "seeded initialisation changed: regenerate tests/golden with make_golden.py"
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"].astype(np.float32))
    xs, ys = synthetic_volume(1, 1, 32, 2, seed=0)
    assert torch.equal(xs, x) and torch.equal(ys, y)
    enc4, logits = m(x.double())
    d, c = oracle_dice_ce_terms(logits, y.double())
    (d + c).backward()
    assert np.allclose(logits.detach()[0, :, ::4, ::4, ::4].numpy(), g["logits_sub"], rtol=1e-5, atol=1e-6)
    assert np.allclose(enc4.detach()[0, ::16, ::2, ::2, ::2].numpy(), g["enc4_sub"], rtol=1e-5, atol=1e-6)
    assert abs(d.item() - float(g["dice"])) < 1e-5 and abs(c.item() - float(g["ce"])) < 1e-5
    gr = dict(m.named_parameters())
    for k in g.files:
        if k.startswith("grad:"):
            ref = g[k]
            got = gr[k[5:]].grad.flatten()[::7].numpy()
            assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-9, k


RANK_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ranking_ref.npz")


def ranking_cases():
    """(key, feat, axis, kind, init_idx, T, loss_f64, loss_f32, grad) from tests/golden/ranking_ref.npz -- outputs of the
    REFERENCE's own extract_triplets_more_partitions / BTLoss / ContrastiveLoss (unetr_ranking_pretraining_3d.py:59-133,
    202-236), executed by tests/golden/make_ranking_golden.py in the build container"""
    z = np.load(RANK_GOLD)
    T = float(z["temperature"])
    for si in range(len(z["shapes"])):
        feat = torch.from_numpy(z[f"s{si}_feat"])
        for axis in (2, 3, 4):
            for kind in ("ranking", "contrastive"):
                k = f"s{si}_ax{axis}_{kind}"
                yield (k, feat, axis, kind, int(z[k + "_init_idx"]), T, float(z[k + "_loss_f64"]), float(z[k + "_loss_f32"]),
                       torch.from_numpy(z[k + "_grad"]))


def test_ranking_oracle_pinned_by_reference_outputs():
    """The oracle's triplet construction and both losses reproduce what the reference's own functions returned on the
    same inputs: loss to 1e-10 relative in fp64 (same arithmetic, same summation order), input gradient to fp32
    rounding of the stored fp64 gradient.  This pins oracle_extract_triplets (order and membership of all 576 triplets
    -- a wrong pairing changes the loss), oracle_bt_loss, oracle_contrastive_loss and oracle_cosine_sim_171."""
    n = 0
    for key, feat, axis, kind, init_idx, T, loss64, loss32, grad in ranking_cases():
        f = feat.double().requires_grad_(True)
        f1, f2 = torch.split(f, [2, 2], dim=0)
        r, s, d = oracle_extract_triplets(f1, f2, axis, init_idx)
        assert len(r) == len(s) == len(d) == 576 and r[0].shape == (feat.shape[1], feat.shape[2] ** 2)
        loss = (oracle_bt_loss if kind == "ranking" else oracle_contrastive_loss)(r, s, d, T)
        assert abs(loss.item() - loss64) <= 1e-10 * abs(loss64), key
        assert abs(loss.item() - loss32) <= 2e-5 * abs(loss64), key      # the reference's fp32 run, for scale
        loss.backward()
        err = (f.grad - grad.double()).abs().max().item()
        assert err <= 2e-7 * grad.abs().max().item(), (key, err)
        n += 1
    assert n == 12


def test_ranking_oracle_pinned_at_config4_sizes():
    """The Bradley-Terry cases of tests/golden/ranking_ref_large.npz (the reference's own BTLoss at BASELINE config[4]'s feature
    sizes [4,128,12,12,12] and [4,2,24,24,24]): the oracle reproduces loss (1e-10, fp64) and the stored gradient samples.  (The
    contrastive cases of that file are loss-only and cost the oracle's Python loop minutes: the HIP kernels are held to them in
    tests/test_ops_gpu.py::test_ranking_losses_large_vs_reference_fixture.)"""
    import importlib.util
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(here, "ranking_ref_large.npz"))
    spec = importlib.util.spec_from_file_location("make_ranking_golden", os.path.join(here, "make_ranking_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    T, stride, n = float(z["temperature"]), int(z["grad_stride"]), 0
    for i, (C, S, axis, contrastive, seed) in enumerate(z["cases"].tolist()):
        if contrastive:
            continue
        f = gen.features(C, S, seed, torch.float32).double().requires_grad_(True)
        f1, f2 = torch.split(f, [2, 2], dim=0)
        r, s_, d = oracle_extract_triplets(f1, f2, axis, int(z[f"c{i}_init_idx"]))
        loss = oracle_bt_loss(r, s_, d, T)
        assert abs(loss.item() - float(z[f"c{i}_loss"])) <= 1e-10 * abs(float(z[f"c{i}_loss"])), i
        loss.backward()
        got = f.grad.flatten()[::stride].numpy()
        assert np.abs(got - z[f"c{i}_grad_sub"]).max() <= 2e-7 * float(z[f"c{i}_grad_absmax"]), i
        n += 1
    assert n == 2


def test_ranking_losses_triplet_structure():
    # unetr_ranking_pretraining_3d.py:59-133: 4 partitions x 12 ordered in-partition pairs x 12 other slices = 576;
    # the reference-pinned value checks are in test_ranking_oracle_pinned_by_reference_outputs above
    torch.manual_seed(0)
    f = torch.randn(4, 8, 12, 12, 12)
    f1, f2 = torch.split(f, [2, 2], dim=0)
    for dim in (2, 3, 4):
        r, s, dsl = oracle_extract_triplets(f1, f2, dim, init_idx=1)
        assert len(r) == len(s) == len(dsl) == 576
        assert r[0].shape == (8, 144)
    bt = oracle_bt_loss(r, s, dsl, 0.1)
    assert torch.isfinite(bt) and bt.item() > 0
    ct = oracle_contrastive_loss(r[:24], s[:24], dsl[:24], 0.5)
    assert torch.isfinite(ct)


def test_oracle_sliding_window_properties():
    """Blending identical per-voxel predictions must reproduce them exactly for every geometry (volume larger than,
    equal to, smaller than the window; odd sizes); the number of predictor calls follows sw_batch_size."""
    from oracle.unetr_oracle import oracle_dice_metric, oracle_post_label, oracle_post_pred, oracle_sliding_window_inference
    calls = []

    def pred(w):
        calls.append(w.shape[0])
        return torch.cat([w * 2 + 1, -w], dim=1)

    for size, roi, ov in ((20, 8, 0.25), (16, 16, 0.25), (6, 8, 0.25), (19, 8, 0.8), (9, 8, 0.0)):
        x = torch.randn(2, 1, size, size, size)
        calls.clear()
        out = oracle_sliding_window_inference(x, (roi,) * 3, 4, pred, overlap=ov)
        assert out.shape == (2, 2, size, size, size)
        assert torch.allclose(out[:, :1], x * 2 + 1, atol=1e-5) and torch.allclose(out[:, 1:], -x, atol=1e-5)
        assert all(c <= 4 for c in calls) and sum(calls) % 2 == 0
    # Dice metric: perfect prediction -> 1 for present classes, NaN (dropped) for absent ones
    y = torch.zeros(2, 1, 4, 4, 4); y[0, 0, :2] = 1
    oh = oracle_post_label(y, 3)
    raw, val = oracle_dice_metric(oh, oh, "mean")
    assert torch.isnan(raw[:, 2]).all() and torch.isnan(raw[1, 1]) and val.item() == 1.0
    logits = torch.randn(2, 3, 4, 4, 4)
    assert torch.equal(oracle_post_pred(logits, 3).argmax(1), logits.argmax(1))


def test_host_window_geometry_matches_oracle(pkg):
    """the host-side window enumeration of inference.sliding_window_inference == the oracle's dense_patch_slices order"""
    import math
    inf = pkg.inference
    for size, roi, ov in ((48, 32, 0.25), (40, 32, 0.5), (96, 96, 0.25), (100, 96, 0.25), (33, 32, 0.8), (160, 96, 0.25)):
        image = [size] * 3
        interval = inf._scan_interval(image, [roi] * 3, ov)
        starts = inf._dense_patch_starts(image, [roi] * 3, interval)
        exp_1d = []
        s = int(roi * (1 - ov)) if roi != size else roi
        n = int(math.ceil(size / s))
        first = next(d for d in range(n) if d * s + roi >= size)
        for i in range(first + 1):
            st = i * s
            exp_1d.append(st - max(st + roi - size, 0))
        assert starts == [(z, y, x) for z in exp_1d for y in exp_1d for x in exp_1d]
