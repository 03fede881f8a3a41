"""Live roofline measurement for bench.py.

Times the hot kernels of training steps with HIP events recorded on the stream the kernels are launched on
(torch's current stream -- the C ABI receives exactly that stream), groups launches by (kernel family,
shape), picks the group with the largest total time (the dominant kernel) and prices it:

  algorithmic bytes  = input feature map read once + output written once + weights (fp32 storage)
  algorithmic flops  = 2 * voxels * Cout * Cin * 27
  bound              = "mfma" if flops/bytes exceeds the ridge (peak_flops / peak_bw) else "hbm"

Peaks from /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s; dense MFMA 2.5 PFLOP/s bf16, 157.3 TFLOP/s
fp32.  `traffic` (PMC HBM bytes) is collected offline with rocprofv3 --pmc (profiles/), null here.
"""
import collections

import torch

HBM_PEAK_GBS = 8000.0
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}


def dominant_kernel_roofline(pkg, model, crit, x, y, precision, reps=3):
    Fn = pkg.functional
    pending = []

    def wrap(name, fam_fn):
        orig = getattr(Fn, name)

        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **k)
            e1.record()
            pending.append((e0, e1) + fam_fn(*a, **k))
            return r
        setattr(Fn, name, timed)
        return orig

    def fam_conv3(xt, ldx, w, dims, prec, mode=0, out=None, ldo=None, accumulate=False):
        B, D, H, W = dims
        v = B * D * H * W
        cout_w, cin_w = w.shape[0], w.shape[1]
        cin, cout = (cin_w, cout_w) if mode == 0 else (cout_w, cin_w)
        nbytes = 4.0 * v * (cin + cout) + 4.0 * w.numel() + (4.0 * v * cout if accumulate else 0.0)
        return ("conv3_fwd_pipe_kernel (3x3x3 LDS-halo implicit GEMM, " + ("dgrad" if mode else "fwd") + ")",
                f"{cin}->{cout} ch @ {D}x{H}x{W}, B={B}", 2.0 * v * cin * cout * 27, nbytes)

    def fam_fused(xt, ldx, w, w3, dims, prec):
        B, D, H, W = dims
        v = B * D * H * W
        cout, cin = w.shape[0], w.shape[1]
        n3 = 1 if w3 is not None else 0
        nbytes = 4.0 * v * (cin + cout * (1 + n3)) + 4.0 * w.numel() + (4.0 * w3.numel() if n3 else 0.0)
        return ("conv3_fwd_pipe_kernel (3x3x3 LDS-halo implicit GEMM, fwd + InstanceNorm sums" + (" + 1x1x1 conv)" if n3 else ")"),
                f"{cin}->{cout} ch @ {D}x{H}x{W}, B={B}", 2.0 * v * cin * cout * (27 + n3), nbytes)

    def fam_wgrad(xt, ldx, dy, lddy, dims, cin, cout, prec, out=None, dy3=None, out3=None):
        B, D, H, W = dims
        v = B * D * H * W
        return ("conv3_wgrad_kernel (+reduce)", f"{cin}->{cout} ch @ {D}x{H}x{W}, B={B}", 2.0 * v * cin * cout * 27,
                4.0 * v * (cin + cout) + 4.0 * cin * cout * 27)

    saved = {"conv3": wrap("conv3", fam_conv3), "conv3_wgrad": wrap("conv3_wgrad", fam_wgrad), "conv3_fused": wrap("conv3_fused", fam_fused)}
    try:
        for _ in range(reps):
            loss = crit(model(x), y)
            loss.backward()
            model.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
    finally:
        for k, v in saved.items():
            setattr(Fn, k, v)
    groups = collections.defaultdict(list)
    for e0, e1, fam, label, flops, nbytes in pending:
        groups[(fam, label)].append((e0.elapsed_time(e1), flops, nbytes))
    tot = {k: sum(t for t, _, _ in v) for k, v in groups.items()}
    (fam, label), _ = max(tot.items(), key=lambda kv: kv[1])
    rows = groups[(fam, label)]
    avg_ms = sum(t for t, _, _ in rows) / len(rows)
    flops, nbytes = rows[0][1], rows[0][2]
    peak_tf = MFMA_PEAK_TFLOPS[precision]
    ridge = peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
    if flops / nbytes > ridge:
        bound, achieved, peak, unit = "mfma", flops / (avg_ms * 1e-3) / 1e12, peak_tf, "TFLOP/s"
    else:
        bound, achieved, peak, unit = "hbm", nbytes / (avg_ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
    return {"kernel": fam, "shape": label, "bound": bound, "achieved": round(achieved, 2), "peak": peak, "unit": unit,
            "frac": round(achieved / peak, 5), "traffic": _pmc_traffic(fam, label, precision), "avg_ms_per_launch": round(avg_ms, 4),
            "launches_per_step": len(rows) // reps, "algorithmic_bytes_per_launch": nbytes,
            "algorithmic_flops_per_launch": flops,
            "share_of_conv_time": round(tot[(fam, label)] / max(sum(tot.values()), 1e-9), 4)}


def _pmc_traffic(fam, label, precision):
    """HBM bytes per launch of the dominant kernel from a committed rocprofv3 --pmc summary (tools/pmc_conv3.sh:
    FETCH_SIZE x 2 (gfx950 half-count of 16 B/lane streams) + WRITE_SIZE, separate passes) whose "family" / "shape"
    keys name the same kernel group; otherwise null."""
    import glob
    import json
    import os
    if precision != "bf16":
        return None
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    for path in sorted(glob.glob(os.path.join(root, "r*_pmc_*.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("family") == fam and d.get("shape") and label.startswith(d["shape"]):
            return d.get("traffic_bytes_per_launch")
    return None


def encoder_forward_rate(pkg, model, x_in, precision, iters=10):
    """ViT-encoder forward (patch embedding + 12 transformer blocks + final LayerNorm) alone, hipGraph replay, as a
    fraction of the dense MFMA peak of the compute dtype (BASELINE.json: 'encoder %MFMA-peak').  Algorithmic FLOPs:
    39.771 GF per 96^3 volume."""
    Fn = pkg.functional
    prec = {"fp32": 0, "bf16": 1}[precision]
    pe = model.vit.patch_embedding
    B = x_in.shape[0]
    L = pe.position_embeddings.shape[1]

    def fwd():
        with torch.no_grad():
            z = Fn.PatchEmbedFn.apply(x_in, pe.patch_embeddings[1].weight, pe.patch_embeddings[1].bias, pe.position_embeddings,
                                      model.patch_size[0], prec)
            for blk in model.vit.blocks:
                z = Fn.TransformerBlockFn.apply(
                    z, blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.out_proj.weight, blk.attn.out_proj.bias,
                    blk.norm2.weight, blk.norm2.bias, blk.mlp.linear1.weight, blk.mlp.linear1.bias, blk.mlp.linear2.weight,
                    blk.mlp.linear2.bias, B, L, model.num_heads, prec)
            return Fn.LayerNormFn.apply(z, model.vit.norm.weight, model.vit.norm.bias)
    fwd()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fwd()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 39.771e9 * B
    tf = flops / (ms * 1e-3) / 1e12
    peak = MFMA_PEAK_TFLOPS[precision]
    return {"batch": B, "ms": round(ms, 4), "TFLOP/s": round(tf, 2), "peak_TFLOP/s": peak, "frac_of_mfma_peak": round(tf / peak, 5),
            "algorithmic_flops": flops,
            "note": (f"batch {B} = {B * L} token rows per GEMM: launch/latency-bound, not MFMA-bound" if B * L < 2048 else
                     f"batch {B} = {B * L} token rows per GEMM")}


def step_report(pkg, model, crit, x, y, precision, ms_per_step):
    """objects bench.py merges into its JSON line"""
    out = {"roofline": dominant_kernel_roofline(pkg, model, crit, x, y, precision)}
    out["encoder_fwd"] = encoder_forward_rate(pkg, model, x, precision)
    for b in (4, 8, 32):
        xb = x[:1].expand(b, -1, -1, -1, -1).contiguous()
        out[f"encoder_fwd_batch{b}"] = encoder_forward_rate(pkg, model, xb, precision, iters=5)
        del xb
    return out
