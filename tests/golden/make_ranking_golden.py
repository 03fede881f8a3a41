"""Generates tests/golden/ranking_ref.npz by EXECUTING the reference's own loss code (build container only).

The three functions `extract_triplets_more_partitions`, `BTLoss` and `ContrastiveLoss` of
/root/reference/unetr_ranking_pretraining_3d.py (lines 59-133, 202-217, 219-236) need nothing but torch, numpy and
itertools, but the module around them imports MONAI (not installed) and does not import as shipped
(`unetr_btcv_segmentation_3d`).  So this script parses the file, takes exactly those three function
definitions out of the syntax tree, and executes them -- unmodified -- in a namespace holding the module-level names
they use (`np`, `torch`, `product`, `permutations`, `num_partitions = 4` (:330), `temperature` (:312/:327),
`cos = CosineSimilarity(dim=-1, eps=1e-6)` (:467)).  `optimizer` is a stub whose step()/zero_grad() do nothing, so
`loss.backward()` inside BTLoss / ContrastiveLoss leaves d loss / d features on the input leaf.

What is committed is DATA only: seeded inputs, the `init_idx` numpy drew, the loss value the reference returned and
the input gradient it produced, for 3 slice axes x 2 loss kinds x 2 feature shapes, in float64 (tight checker for
the oracle) and float32 (what the reference's GPU run would have seen).  The reference file itself never travels.

torch here is 2.10 (the reference pins 1.7.1): CosineSimilarity's eps handling differs only for slices of norm
< 1e-6; the synthetic features below have slice norms of O(10).

    python tests/golden/make_ranking_golden.py        # writes tests/golden/ranking_ref.npz
"""
import ast
import contextlib
import io
import os
import re
import sys
from itertools import permutations, product

import numpy as np
import torch
from torch.nn import CosineSimilarity

REF = "/root/reference/unetr_ranking_pretraining_3d.py"
WANTED = ("extract_triplets_more_partitions", "BTLoss", "ContrastiveLoss")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ranking_ref.npz")

# (C, S) of the [4, C, S, S, S] feature batch: a "feat"-stage-like map (many channels, small grid; enc4 is
# [4,128,12,12,12] at 96^3) and a "recon"-stage-like map (few channels = classes, larger grid)
SHAPES = ((4, 8), (2, 12))
TEMPERATURE = 0.1          # the value of every usage line in the reference docstring (:302-305)


class _NoOpOptimizer:
    def step(self):
        pass

    def zero_grad(self):
        pass


def load_reference_functions(temperature):
    with open(REF) as f:
        tree = ast.parse(f.read(), REF)
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert sorted(d.name for d in defs) == sorted(WANTED), [d.name for d in defs]
    ns = {"np": np, "torch": torch, "product": product, "permutations": permutations, "num_partitions": 4,
          "temperature": temperature, "cos": CosineSimilarity(dim=-1, eps=1e-6)}
    exec(compile(ast.Module(body=defs, type_ignores=[]), REF, "exec"), ns)
    return ns


def features(C, S, seed, dtype):
    """smooth + noisy positive-ish feature maps so that slices of one partition correlate more than distant ones"""
    g = torch.Generator().manual_seed(seed)
    ax = torch.linspace(-1, 1, S, dtype=torch.float64)
    zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
    f = torch.randn(4, C, S, S, S, generator=g, dtype=torch.float64) * 0.5
    for b in range(4):
        for c in range(C):
            ph = torch.rand(3, generator=g, dtype=torch.float64) * 3.0
            f[b, c] += torch.sin(2.0 * zz + ph[0]) + torch.cos(1.5 * yy + ph[1]) * torch.sin(xx + ph[2]) + 0.3 * (b % 2)
    return f.to(dtype)


def main():
    if not os.path.exists(REF):
        raise SystemExit("the reference is only present in the build container")
    ns = load_reference_functions(TEMPERATURE)
    out = {"temperature": np.float64(TEMPERATURE), "shapes": np.array(SHAPES, dtype=np.int64)}
    n = 0
    for si, (C, S) in enumerate(SHAPES):
        base = features(C, S, 7 + si, torch.float32)       # fp32-representable values: both runs below see the same numbers
        out[f"s{si}_feat"] = base.numpy()
        for axis in (2, 3, 4):
            for kind, fn_name in (("ranking", "BTLoss"), ("contrastive", "ContrastiveLoss")):
                seed = 100 * si + 10 * axis + (kind == "contrastive")
                key = f"s{si}_ax{axis}_{kind}"
                for dt_name, dtype in (("f64", torch.float64), ("f32", torch.float32)):
                    feat = base.detach().clone().to(dtype).requires_grad_(True)
                    f1, f2 = torch.split(feat, [2, 2], dim=0)          # :264
                    part = int(S / 4)
                    np.random.seed(seed)
                    init_idx = int(np.random.choice(np.arange(0, part)))   # the draw the function is about to make (:75)
                    np.random.seed(seed)
                    sink = io.StringIO()
                    with contextlib.redirect_stdout(sink):               # the reference prints every slice shape
                        ref, sim, dis = ns["extract_triplets_more_partitions"](f1, f2, axis)
                        loss = ns[fn_name](ref, sim, dis, _NoOpOptimizer())
                    line = [l for l in sink.getvalue().splitlines() if l.startswith("Slice indices:")][0]
                    assert [int(v) for v in re.findall(r"\d+", line.replace("int64", ""))] == [init_idx + p * part for p in range(4)], line
                    assert len(ref) == len(sim) == len(dis) == 576
                    out[f"{key}_loss_{dt_name}"] = np.float64(loss)
                    if dt_name == "f64":                                 # fp64 run's gradient, stored rounded to fp32
                        out[key + "_grad"] = feat.grad.numpy().astype(np.float32)
                    n += 1
                out[key + "_init_idx"] = np.int64(init_idx)
    np.savez_compressed(OUT, **out)
    print(f"{n} reference runs -> {OUT} ({os.path.getsize(OUT) / 1024:.0f} KiB)")
    large(ns)


# The two feature sizes of BASELINE config[4] at 96^3 -- enc4 [4,128,12,12,12] ('feat' stage) and the logits-like [4,2,24,24,24]
# -- through the reference's BTLoss / ContrastiveLoss.  The inputs are NOT stored: they are features(C, S, seed, float32) of this
# file, which the test regenerates (torch's CPU generator is deterministic).  BTLoss (576 terms): loss and every 97th element of
# the input gradient.  ContrastiveLoss (576 x 577 cosine terms): its autograd graph at these sizes takes > 40 GB (the run was
# killed at 37 GB), so it is executed WITHOUT a graph -- torch.no_grad() and Tensor.backward patched to a no-op around the
# reference's unmodified function -- and only the loss value is stored.
LARGE = ((128, 12, 2, "ranking", 41), (128, 12, 3, "contrastive", 44), (2, 24, 4, "ranking", 42), (2, 24, 3, "contrastive", 43))
OUT_LARGE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ranking_ref_large.npz")
GRAD_STRIDE = 97


def large(ns):
    out = {"temperature": np.float64(TEMPERATURE), "grad_stride": np.int64(GRAD_STRIDE),
           "cases": np.array([(C, S, ax, int(k == "contrastive"), seed) for C, S, ax, k, seed in LARGE], dtype=np.int64)}
    for i, (C, S, axis, kind, seed) in enumerate(LARGE):
        with_grad = kind == "ranking"
        feat = features(C, S, seed, torch.float32).to(torch.float64).requires_grad_(with_grad)
        f1, f2 = torch.split(feat, [2, 2], dim=0)
        part = int(S / 4)
        np.random.seed(seed)
        init_idx = int(np.random.choice(np.arange(0, part)))
        np.random.seed(seed)
        saved_backward = torch.Tensor.backward
        try:
            if not with_grad:
                torch.Tensor.backward = lambda self, *a, **k: None
            with contextlib.redirect_stdout(io.StringIO()), torch.set_grad_enabled(with_grad):
                ref, sim, dis = ns["extract_triplets_more_partitions"](f1, f2, axis)
                loss = ns["ContrastiveLoss" if kind == "contrastive" else "BTLoss"](ref, sim, dis, _NoOpOptimizer())
        finally:
            torch.Tensor.backward = saved_backward
        out[f"c{i}_loss"] = np.float64(loss)
        out[f"c{i}_init_idx"] = np.int64(init_idx)
        if with_grad:
            g = feat.grad.flatten()
            out[f"c{i}_grad_sub"] = g[::GRAD_STRIDE].numpy().astype(np.float32)
            out[f"c{i}_grad_absmax"] = np.float64(g.abs().max())
        print(f"large case {i}: C={C} S={S} axis={axis} {kind}: loss {float(loss):.6f}", flush=True)
    np.savez_compressed(OUT_LARGE, **out)
    print(f"-> {OUT_LARGE} ({os.path.getsize(OUT_LARGE) / 1024:.0f} KiB)")


if __name__ == "__main__":
    sys.exit(main())
