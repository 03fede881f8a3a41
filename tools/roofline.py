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
        return ("conv3_fwd_kernel (3x3x3 LDS-halo implicit GEMM, " + ("dgrad" if mode else "fwd") + ")",
                f"{cin}->{cout} ch @ {D}x{H}x{W}, B={B}", 2.0 * v * cin * cout * 27, nbytes)

    def fam_wgrad(xt, ldx, dy, lddy, dims, cin, cout, prec, out=None, dy3=None, out3=None):
        B, D, H, W = dims
        v = B * D * H * W
        return ("conv3_wgrad_kernel (+reduce)", f"{cin}->{cout} ch @ {D}x{H}x{W}, B={B}", 2.0 * v * cin * cout * 27,
                4.0 * v * (cin + cout) + 4.0 * cin * cout * 27)

    saved = {"conv3": wrap("conv3", fam_conv3), "conv3_wgrad": wrap("conv3_wgrad", fam_wgrad)}
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
            "frac": round(achieved / peak, 5), "traffic": None, "avg_ms_per_launch": round(avg_ms, 4),
            "launches_per_step": len(rows) // reps, "algorithmic_bytes_per_launch": nbytes,
            "algorithmic_flops_per_launch": flops,
            "share_of_conv_time": round(tot[(fam, label)] / max(sum(tot.values()), 1e-9), 4)}
