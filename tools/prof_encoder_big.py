"""ViT-encoder forward alone at a many-row batch (PROBE_B, default 32 = 6912 token rows), replayed from a hipGraph -- the run that
`rocprofv3 --kernel-trace --stats -- python3 tools/prof_encoder_big.py` breaks down per kernel (what bench.py reports as
encoder_fwd_batch32).  Prints ms per forward and the fraction of the dense bf16 MFMA peak."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
from tools import roofline as rl  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B = int(os.environ.get("PROBE_B", 32))
    torch.manual_seed(0)
    model = pkg.UNETR(in_channels=1, out_channels=4, img_size=(96, 96, 96), feature_size=16, hidden_size=768, mlp_dim=3072,
                      num_heads=12, pos_embed="perceptron", norm_name="instance", res_block=True).to(dev)
    model.precision = "bf16"
    x = torch.randn(B, 1, 96, 96, 96, device=dev)
    # as in the bench run, where this package's AdamW keeps the bf16 weight shadows in step: one eager pass creates them, then they
    # count as optimizer-maintained (otherwise every captured forward re-casts its 49 weights)
    rl.encoder_forward_rate(pkg, model, x[:1], "bf16", iters=1)
    for ent in pkg.functional._SHADOW.values():
        ent[3] = True
    r = rl.encoder_forward_rate(pkg, model, x, "bf16", iters=int(os.environ.get("PROBE_ITERS", 10)))
    print(r, flush=True)


if __name__ == "__main__":
    main()
