"""Whole-step kernel breakdown and roofline for bench.py (config[1]: 96^3, hidden 768, batch B per GPU, bf16 mode).

How the numbers are made (all inside the bench process, on the GPU that ran the step):

* per-kernel device time: torch.profiler (roctracer) around replays of the captured step -- the same hipGraph the timed
  region replays -- summed per kernel name, divided by the number of replays;
* kernels are grouped into FAMILIES (regex on the kernel name); every family has an analytic model of its ALGORITHMIC work
  per step (bytes it must move, flops it must do -- tensor sizes of the reference architecture; nothing
  measured goes into it) and the roofline that bounds it: HBM (8 TB/s) for streaming / low-intensity kernels, bf16 MFMA
  (2.5 PFLOP/s dense) where flops / bytes exceeds the ridge;
* `roofline` = the family with the largest time per step: achieved = algorithmic work / measured time, frac = achieved / peak,
  `traffic` = HBM bytes per launch from the committed rocprofv3 --pmc summary named in `traffic_source` (FETCH_SIZE x 2 x 1024
  + WRITE_SIZE x 1024, separate passes, gfx950 correction of MI355X_MICROARCH.md) or null;
* `families` = the same for every family (ms per step, launches, achieved, frac), so the line shows where the step goes.
"""
import collections
import json
import os
import re

import torch

HBM_PEAK_GBS = 8000.0
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}

# (family, regex on the demangled kernel name)
FAMILIES = [
    ("AdamW (fp32 master + moments, bf16 shadow)", r"adamw_kernel|adamw_ranges_kernel"),
    ("ViT Linear GEMMs fwd + data-grad (bf16 operands)", r"gemm_bf16_kernel|gemm_bf16_big_kernel|splitk_reduce_kernel<.*EpBf"),
    ("ViT Linear weight-grad (grouped)", r"gemm_bf16_grouped_wgrad|gemm_grouped_wgrad"),
    ("attention fwd + bwd", r"attn16_|attn_fwd|attn_bwd"),
    ("LayerNorm fwd + bwd", r"layernorm_"),
    ("InstanceNorm passes", r"in_apply|in_bwd_|in_stats"),
    ("3x3x3 conv fwd + data-grad", r"conv3_fwd|conv3_c1_fwd"),
    ("3x3x3 conv weight-grad", r"conv3_wgrad|conv3_c1_wgrad|reduce_rows_grouped"),
    ("2x2x2 transposed conv (voxel-tile kernels, shuffles)", r"tconv2_|pixel_"),
    ("out conv + DiceCE", r"outconv|dicece"),
]


def _arch(B, fused_update=False):
    """algorithmic work per step of config[1] at batch B (bytes with bf16 feature maps, flops = 2 MAC).  fused_update: AdamW of the
    88.3 M ViT Linear weights runs in the epilogue of the grouped weight-gradient launch (26 bytes per weight there -- masters and
    moments read + written, bf16 shadow written -- instead of a 4-byte gradient store plus 30 bytes in the optimizer kernel)"""
    P = 92452868
    res = [(96 ** 3, 1, 16), (12 ** 3, 256, 128), (24 ** 3, 128, 64), (48 ** 3, 64, 32), (96 ** 3, 32, 16)]   # (V, Cin, Cout) of the 5 UnetResBlocks
    a = {}
    a["AdamW (fp32 master + moments, bf16 shadow)"] = dict(bytes=30.0 * (P - 88.3e6 if fused_update else P), flops=0.0)
    lin_fwd = B * 1e9 * (1.359 + 12 * (0.7644 + 0.2548 + 2.0384))
    lin_dg = B * 1e9 * 12 * (0.7644 + 0.2548 + 2.0384)
    # per GEMM: A operand + weight + every output it stores (and the residual / GELU' argument its epilogue reads): what the launch
    # must move when every operand is read once
    M = 216.0 * B
    H, F, PD = 768.0, 3072.0, 4096.0
    blk_fwd = ((M * H * 2 + 3 * H * H * 2 + M * 3 * H * 2)                 # qkv: x bf16, W, qkv bf16 out
               + (M * H * 2 + H * H * 2 + M * H * 4 * 2)                   # out-proj: attn bf16, W, fp32 out + residual read
               + (M * H * 2 + F * H * 2 + M * F * (4 + 2))                 # linear1: x bf16, W, fp32 pre-activation + bf16 GELU out
               + (M * F * 2 + H * F * 2 + M * H * 4 * 2))                  # linear2: h bf16, W, fp32 out + residual read
    blk_dg = ((M * H * 2 + H * F * 2 + M * F * (4 + 2))                    # d linear2: dy bf16, W, GELU' argument read + bf16 out
              + (M * F * 2 + F * H * 2 + M * H * 4)                        # d linear1: du bf16, W, fp32 out
              + (M * H * 2 + H * H * 2 + M * H * 2)                        # d out-proj: dy bf16, W, bf16 out
              + (M * 3 * H * 2 + 3 * H * H * 2 + M * H * 4))               # d qkv: dqkv bf16, W, fp32 out
    pe = M * PD * 2 + H * PD * 2 + M * H * 4 * 2
    a["ViT Linear GEMMs fwd + data-grad (bf16 operands)"] = dict(bytes=12 * (blk_fwd + blk_dg) + pe, flops=lin_fwd + lin_dg)
    a["ViT Linear weight-grad (grouped)"] = dict(bytes=(26.0 if fused_update else 4.0) * 88.3e6, flops=lin_fwd)
    a["attention fwd + bwd"] = dict(bytes=B * 216 * 768 * 2.0 * 12 * (4 + 7), flops=B * 1e9 * 12 * 0.1434 * 3.5)
    a["LayerNorm fwd + bwd"] = dict(bytes=B * 216 * 768 * (25 * (4 + 2) + 25 * (4 * 3 + 4 + 2)), flops=0.0)
    inb = cvb = cwb = 0.0
    cvf = cwf = 0.0
    for V, ci, co in res:
        u = 2.0 * B * V * co                     # one [V, Cout] bf16 feature map
        # fwd: apply (r1 w1), dual apply (r2 w1); bwd: dual reduce r3, dual apply r3 w2, single apply r2 w1 (the single form's reduction
        # rides in the epilogue of the data-gradient conv since round 4: its extra read of c1 is counted there)
        inb += 16 * u
        xin = (4.0 if ci == 1 else 2.0) * B * V * ci
        # fwd: conv1 (+1x1): read x, write c1, c3; conv2: read a1, write c2.  dgrad: conv2^T: r dc2 w da1; block input grad: r dc1, dc3, w dx
        cvb += xin + 2 * u + 2 * u + 3 * u + (0 if ci == 1 else 2 * u + xin)
        cvf += 2.0 * B * V * co * (27 * ci + ci + 27 * co) + 2.0 * B * V * 27 * co * co + (0 if ci == 1 else 2.0 * B * V * ci * co * 28)
        cwb += (xin + 2 * u) + 2 * u             # conv1 wgrad (+1x1): x, dc1, dc3; conv2 wgrad: a1, dc2
        cwf += 2.0 * B * V * co * (28 * ci + 27 * co)
    a["InstanceNorm passes"] = dict(bytes=inb, flops=0.0)
    a["3x3x3 conv fwd + data-grad"] = dict(bytes=cvb, flops=cvf)
    a["3x3x3 conv weight-grad"] = dict(bytes=cwb, flops=cwf)
    tc = [(6 ** 3, 768, 32), (12 ** 3, 32, 32), (24 ** 3, 32, 32), (6 ** 3, 768, 64), (12 ** 3, 64, 64), (6 ** 3, 768, 128),
          (6 ** 3, 768, 128), (12 ** 3, 128, 64), (24 ** 3, 64, 32), (48 ** 3, 32, 16)]                 # (V_in, Cin, Cout) of the 10 transposed convs
    tb = sum(2.0 * B * V * (ci + 8 * co) for V, ci, co in tc) * 3        # fwd, dgrad, wgrad each touch input + output once
    tf = sum(2.0 * B * V * ci * co * 8 for V, ci, co in tc) * 3
    a["2x2x2 transposed conv (voxel-tile kernels, shuffles)"] = dict(bytes=tb, flops=tf)
    a["out conv + DiceCE"] = dict(bytes=B * 96 ** 3 * (2.0 * 16 * 3 + 4.0 * 4 * 5 + 4.0 * 2), flops=0.0)
    return a


def _family_of(name):
    for fam, rx in FAMILIES:
        if re.search(rx, name):
            return fam
    return "other (packs, casts, column sums, copies, fills)"


def _kernel_times(run, reps=7, once_per_step="dicece_fwd_kernel"):
    """{kernel name: (device us per step, launches per step)} of `run()` via torch.profiler.  The tracer may drop records of the
    replays at either end of the window, so only the launches between the first and the last occurrence of a kernel that runs
    exactly once per step (the loss forward) are counted: whole steps, mid-step to mid-step."""
    from torch.profiler import ProfilerActivity, profile
    run()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
    raw = []
    for ev in prof.events():
        if "cuda" not in str(getattr(ev, "device_type", "")).lower():
            continue
        k = ev.name
        if k.startswith("hip") or k.startswith("Memcpy") or k.startswith("Memset"):
            continue
        tr = ev.time_range
        raw.append((tr.start, tr.end - tr.start, k))
    raw.sort()
    marks = [i for i, (_, _, k) in enumerate(raw) if once_per_step in k]
    if len(marks) >= 2:
        raw, steps = raw[marks[0]:marks[-1]], len(marks) - 1
    else:
        steps = reps
    acc = {}
    for _, d, k in raw:
        t, c = acc.get(k, (0.0, 0))
        acc[k] = (t + d, c + 1)
    return {k: (t / steps, c / steps) for k, (t, c) in acc.items() if t > 0}


def _short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    cut = min([i for i in (n.find("("),) if i > 0] or [len(n)])
    return n[:cut][:90]


def _pmc_traffic(kernel_rx):
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    d, rel = None, None
    import glob
    for cand in sorted((os.path.basename(f) for f in glob.glob(os.path.join(root, "r[0-9][0-9]_pmc_step_kernels.json"))), reverse=True):   # newest round first
        try:
            d, rel = json.load(open(os.path.join(root, cand))), "profiles/" + cand
            break
        except (OSError, ValueError):
            continue
    if d is None:
        return None, None
    tot, n = 0.0, 0
    for k, v in d.get("kernels", {}).items():
        if re.search(kernel_rx, k):
            tot += v["hbm_bytes_per_step"]
            n += v["launches_per_step"]
    if n == 0:
        return None, None
    return tot / n, rel + " (" + d.get("command", "rocprofv3 --pmc") + ")"


def step_report(pkg, step, batch, precision, ms_per_step):
    times = _kernel_times(step.run)
    arch = _arch(batch, fused_update=any(re.search(r"grouped_wgrad_kernel<\d+, *\d+, *(1|true)>", k) for k in times))
    fam = collections.OrderedDict()
    for name, (us, n) in times.items():
        f = _family_of(name)
        e = fam.setdefault(f, {"ms": 0.0, "launches": 0.0, "kernels": collections.Counter()})
        e["ms"] += us / 1e3
        e["launches"] += n
        e["kernels"][_short(name)] += us / 1e3
    peak_tf = MFMA_PEAK_TFLOPS[precision]
    ridge = peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
    table = []
    for f, e in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
        row = {"family": f, "ms_per_step": round(e["ms"], 4), "launches_per_step": round(e["launches"], 1),
               "top_kernel": max(e["kernels"].items(), key=lambda kv: kv[1])[0]}
        w = arch.get(f)
        if w is not None and e["ms"] > 0:
            sec = e["ms"] * 1e-3
            f_hbm = w["bytes"] / sec / 1e9 / HBM_PEAK_GBS
            f_mfma = w["flops"] / sec / 1e12 / peak_tf
            if w["flops"] / max(w["bytes"], 1.0) > ridge:
                row.update(bound="mfma", achieved=round(w["flops"] / sec / 1e12, 2), peak=peak_tf, unit="TFLOP/s")
            else:
                row.update(bound="hbm", achieved=round(w["bytes"] / sec / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s")
            row["frac"] = round(row["achieved"] / row["peak"], 4)
            row["frac_of_hbm_peak"], row["frac_of_mfma_peak"] = round(f_hbm, 4), round(f_mfma, 4)
            row["algorithmic_bytes_per_step"] = w["bytes"]
            row["algorithmic_flops_per_step"] = w["flops"]
        table.append(row)
    out = {"families": table, "kernel_time_ms_per_step": round(sum(r["ms_per_step"] for r in table), 4),
           "kernel_time_note": "profiler-side sum of kernel durations over profiled replays: it may exceed ms_per_step (un-profiled "
                               "wall clock of the timed region) by about 1 %"}
    dom = next((r for r in table if "bound" in r), None)
    if dom is not None:
        rx = dict(FAMILIES)[dom["family"]]
        traffic, src = _pmc_traffic(rx)
        n = max(dom["launches_per_step"], 1.0)
        per = "bytes" if dom["bound"] == "hbm" else "flops"
        out["roofline"] = {
            "kernel": dom["family"] + ": " + dom["top_kernel"], "bound": dom["bound"], "achieved": dom["achieved"], "peak": dom["peak"],
            "unit": dom["unit"], "frac": dom["frac"], "traffic": traffic, "traffic_source": src,
            "avg_ms_per_launch": round(dom["ms_per_step"] / n, 5), "launches_per_step": dom["launches_per_step"],
            "algorithmic_%s_per_launch" % per: dom["algorithmic_%s_per_step" % per] / n,
            "share_of_step_kernel_time": round(dom["ms_per_step"] / max(out["kernel_time_ms_per_step"], 1e-9), 4),
            "how": "device time of the family's kernels per replayed step (torch.profiler over hipGraph replays) against its analytic "
                   "algorithmic work (tools/roofline.py:_arch)"}
    return out


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
            blocks = list(model.vit.blocks)
            for i, blk in enumerate(blocks):           # (as UNETR._encode: the next block's norm1 rides on this block's last kernel)
                nxt = blocks[i + 1].norm1 if i + 1 < len(blocks) else None
                z = Fn.TransformerBlockFn.apply(
                    z, blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.out_proj.weight, blk.attn.out_proj.bias,
                    blk.norm2.weight, blk.norm2.bias, blk.mlp.linear1.weight, blk.mlp.linear1.bias, blk.mlp.linear2.weight,
                    blk.mlp.linear2.bias, B, L, model.num_heads, prec, False,
                    None if nxt is None else nxt.weight.detach(), None if nxt is None else nxt.bias.detach())
            return Fn.LayerNormFn.apply(z, model.vit.norm.weight, model.vit.norm.bias)
    fwd()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
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
            "note": (f"batch {B} = {B * L} token rows per GEMM: 61 dependent launches, each bounded by launch boundary + the ~70 GB/s a "
                     f"CU pulls from L2, not by MFMA rate (DESIGN.md section 5)" if B * L < 2048 else f"batch {B} = {B * L} token rows per GEMM")}


def encoder_report(pkg, model, x, precision):
    out = {"encoder_fwd": encoder_forward_rate(pkg, model, x, precision)}
    for b in (4, 8, 32):
        xb = x[:1].expand(b, -1, -1, -1, -1).contiguous()
        out[f"encoder_fwd_batch{b}"] = encoder_forward_rate(pkg, model, xb, precision, iters=5)
        del xb
    return out
