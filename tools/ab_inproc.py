"""A/B of environment switches INSIDE one process: one captured training step (config[1], bf16, batch 2, fused optimizer epilogue --
exactly what bench.py times) per setting, then the settings are timed round-robin in short windows with HIP events, so that clock /
thermal drift hits every arm alike.  The switches must be read at call time (Python os.environ or getenv per launch): they are
baked into the graph at capture.

    python tools/ab_inproc.py "UNETR_AMD_IN_FUSE=0" "UNETR_AMD_IN_FUSE=3" "UNETR_AMD_IN_FUSE=3 UNETR_IN_FIN=0" [--rounds 12] [--steps 20]

Prints per arm: median / min ms per step over the rounds, and the per-round differences to the first arm (median).
"""
import argparse
import importlib
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CFG = dict(in_channels=1, out_channels=4, img_size=(96, 96, 96), feature_size=16, hidden_size=768, mlp_dim=3072, num_heads=12,
           pos_embed="perceptron", norm_name="instance", res_block=True)


def build(pkg, dev, env, batch):
    from tools.synthetic import synthetic_volume
    saved = {}
    for kv in env.split():
        k, v = kv.split("=", 1)
        saved[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        torch.manual_seed(1234)
        model = pkg.UNETRLogits(**CFG).to(dev)
        model.precision = "bf16"
        crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)
        flat = model.use_flat_buffers()
        opt = pkg.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, flat=flat)
        x, y = synthetic_volume(batch, 1, 96, 4, seed=1234)
        step = pkg.TrainStep(model, crit, opt, x.to(dev), y.to(dev), use_graph=True, fuse_update=True)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("arms", nargs="+", help='each arm: "VAR=val VAR2=val2" ("-" = no switches)')
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
    dev = torch.device("cuda:0")
    steps = [build(pkg, dev, "" if arm == "-" else arm, a.batch) for arm in a.arms]
    for st in steps:
        for _ in range(5):
            st.run()
    torch.cuda.synchronize()
    times = [[] for _ in steps]
    for r in range(a.rounds):
        order = list(range(len(steps)))
        if r % 2:
            order.reverse()
        for i in order:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            steps[i].run()                      # one untimed replay: caches hold this arm's buffers, not the previous arm's
            e0.record()
            for _ in range(a.steps):
                steps[i].run()
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / a.steps)
    for arm, t, st in zip(a.arms, times, steps):
        d = [x - y for x, y in zip(t, times[0])]
        print(f"{arm:50s} median {statistics.median(t):.4f}  min {min(t):.4f}  vs arm0 {statistics.median(d):+.4f} ms  loss {float(st.loss.item()):.6f}")


if __name__ == "__main__":
    main()
