"""Where the bf16 mode's forward error comes from: configs[1] geometry at batch 1, logits / enc4 / Dice / CE of
every GPU precision mode against the fp32 CPU oracle at the same seed-0 weights and volume.

    python tools/decompose_bf16_error.py                     fp32, bf16x3, bf16 (bf16 storage + bf16 operands)
    UNETR_AMD_LIB=.../libunetr_hip_x3droplo.so python tools/decompose_bf16_error.py x3only
                                                             "bf16x3" of the diagnostic build = bf16 x 1 operands, fp32 storage
"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
from oracle.unetr_oracle import OracleUNETR, oracle_dice_ce_terms, synthetic_volume  # noqa: E402  (diagnostic tool, not the product)

C2 = dict(in_channels=1, out_channels=4, img_size=(96, 96, 96), feature_size=16, hidden_size=768, mlp_dim=3072, num_heads=12,
          pos_embed="perceptron", norm_name="instance", res_block=True)
dev = torch.device("cuda:0")
torch.manual_seed(0)
ref = OracleUNETR(**C2)
x, y = synthetic_volume(1, 1, 96, 4, seed=11)
with torch.no_grad():
    enc4_r, logits_r = ref(x)
    d_r, c_r = oracle_dice_ce_terms(logits_r, y)
rel = lambda a, b: float(((a.detach().float().cpu() - b).abs().max() / b.abs().max()).item())
modes = ["bf16x3"] if sys.argv[1:] == ["x3only"] else ["fp32", "bf16x3", "bf16"]
tag = " (lo halves dropped: bf16 x 1 operands, fp32 storage)" if "x3droplo" in (os.environ.get("UNETR_AMD_LIB") or "") else ""
for mode in modes:
    hip = pkg.UNETR(**C2)
    hip.load_state_dict(ref.state_dict(), strict=True)
    hip = hip.to(dev)
    hip.precision = mode
    with torch.no_grad():
        enc4, logits = hip(x.to(dev))
        t = pkg.DiceCELoss(to_onehot_y=True, softmax=True).terms(logits, y.to(dev))
    print(f"{mode + tag:70s} logits {rel(logits, logits_r):.2e}  enc4 {rel(enc4, enc4_r):.2e}  dice {abs(float(t[1]) - float(d_r)) / abs(float(d_r)):.2e}  "
          f"ce {abs(float(t[2]) - float(c_r)) / abs(float(c_r)):.2e}", flush=True)
    del hip
