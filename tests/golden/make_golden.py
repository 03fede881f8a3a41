"""Generates tests/golden/c1_seed0.npz from the CPU oracle (oracle/unetr_oracle.py).

PARITY UNPINNED: the reference's own arithmetic (MONAI 0.6.0) is not importable here and the reference ships
no fixtures, so these vectors pin the oracle against ITSELF (regression guard) and give the GPU tests a
file-based target that does not need the oracle at run time.  Weights are regenerated from the seed (storing
5.2 M parameters would be 20 MB); inputs, sub-sampled outputs, loss terms and sub-sampled gradients are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.unetr_oracle import OracleUNETR, oracle_dice_ce_terms, synthetic_volume  # noqa: E402

C1 = dict(in_channels=1, out_channels=2, img_size=(32, 32, 32), feature_size=16, hidden_size=128, mlp_dim=512,
          num_heads=4, pos_embed="perceptron", norm_name="instance", res_block=True)
GRAD_KEYS = ["vit.blocks.0.attn.qkv.weight", "vit.blocks.11.mlp.linear2.weight", "encoder1.layer.conv1.conv.weight",
             "decoder2.conv_block.conv2.conv.weight", "out.conv.conv.bias", "vit.norm.weight"]


def build(seed=0):
    torch.manual_seed(seed)
    return OracleUNETR(**C1)


def main():
    torch.set_num_threads(4)
    # fp64 oracle: the fp32 oracle's deep-layer gradients move by ~1e-2 relative with the BLAS thread count
    # (InstanceNorm over 4^3 voxels at the bottleneck is ill-conditioned), fp64 is reproducible to ~1e-12
    m = build(0).double()
    x, y = synthetic_volume(1, 1, 32, 2, seed=0)
    enc4, logits = m(x.double())
    dice, ce = oracle_dice_ce_terms(logits, y.double())
    (dice + ce).backward()
    g = dict(m.named_parameters())
    out = {
        "x": x.numpy(), "y": y.numpy().astype(np.uint8),
        "enc4_sub": enc4.detach()[0, ::16, ::2, ::2, ::2].float().numpy(),
        "logits_sub": logits.detach()[0, :, ::4, ::4, ::4].float().numpy(),
        "enc4_absmax": np.float32(enc4.abs().max().item()), "logits_absmax": np.float32(logits.abs().max().item()),
        "logits_mean": np.float32(logits.mean().item()),
        "dice": np.float32(dice.item()), "ce": np.float32(ce.item()),
        "weight_checksum": np.float64(sum(p.double().sum().item() for p in m.parameters())),
    }
    for k in GRAD_KEYS:
        out["grad:" + k] = g[k].grad.flatten()[::7].float().numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c1_seed0.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
