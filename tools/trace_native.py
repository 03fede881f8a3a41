"""Which torch-native device work (elementwise add / fill / copy kernels, memcpy nodes) is inside ONE eager training step, and
which Python line launches it: torch.profiler with stacks over one eager step of the benched configuration."""
import importlib
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
from tools.synthetic import synthetic_volume  # noqa: E402

dev = torch.device("cuda:0")
CFG = dict(in_channels=1, out_channels=4, img_size=(96, 96, 96), feature_size=16, hidden_size=768, mlp_dim=3072,
           num_heads=12, pos_embed="perceptron", norm_name="instance", res_block=True)
torch.manual_seed(1234)
model = pkg.UNETRLogits(**CFG).to(dev)
model.precision = "bf16"
flat = model.use_flat_buffers()
opt = pkg.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, flat=flat)
crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)
x, y = synthetic_volume(2, 1, 96, 4, seed=1234)
x, y = x.to(dev), y.to(dev)
dp = len(sys.argv) > 1 and sys.argv[1] == "dp"
step = pkg.TrainStep(model, crit, opt, x, y, use_graph=False, data_parallel=dp, warmup=2)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step.run()
    torch.cuda.synchronize()
ev = prof.events()
native = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::")
          and any(k in e.name for k in ("add", "fill", "zero", "copy", "clone", "contiguous", "to", "mul", "empty_strided", "cat", "sum", "item"))]
seen = {}
for e in native:
    if e.cuda_time_total <= 0 and "copy" not in e.name and "fill" not in e.name:
        continue
    stack = [s for s in (e.stack or []) if "3dmedical" in s or "train_step" in s or "optim" in s or "losses" in s]
    key = (e.name, tuple(stack[:3]), str(e.input_shapes)[:80])
    seen.setdefault(key, [0, 0.0])
    seen[key][0] += 1
    seen[key][1] += e.cuda_time_total
for (name, stack, shapes), (n, t) in sorted(seen.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:3d}x {t:8.1f} us  {name}  {shapes}")
    for s in stack:
        print("        ", s)
print("---- device kernels that are not ours")
for e in prof.key_averages():
    if e.device_type == torch.autograd.DeviceType.CUDA and ("at::native" in e.key or "rocclr" in e.key or "Memcpy" in e.key or "Memset" in e.key):
        print(f"{e.count:3d}x {e.device_time_total:8.1f} us  {e.key[:120]}")
