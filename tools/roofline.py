"""Live roofline measurement for bench.py: times the hot kernels of one training step with HIP events on
the stream they are launched on, picks the dominant kernel family and prices it against its roofline.

Algorithmic work per launch comes from the shapes (SURVEY.md 8d / DESIGN.md): FLOPs = 2*M*N*K of the
implicit GEMM for MFMA-bound kernels, bytes = operands read once + result written once for HBM-bound ones.
"""
import collections

import torch

HBM_PEAK_GBS = 8000.0
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}


def dominant_kernel_roofline(pkg, model, crit, x, y, precision, reps=3):
    Fn = pkg.functional
    records = collections.defaultdict(list)   # family -> [(ms, flops, bytes, label)]

    def wrap(name, fam_fn):
        orig = getattr(Fn, name)

        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **k)
            e1.record()
            fam, flops, nbytes, label = fam_fn(*a, **k)
            pending.append((fam, e0, e1, flops, nbytes, label))
            return r
        setattr(Fn, name, timed)
        return orig

    def fam_conv_fwd(xt, ldx, wpack, dims, cin, cout, ks, prec, out=None, ldo=None, accumulate=False):
        B, D, H, W = dims
        v = B * D * H * W
        return (f"conv{ks}x{ks}x{ks}_igemm", 2.0 * v * cout * cin * ks ** 3, 4.0 * v * (cin + cout) + 4.0 * cin * cout * ks ** 3,
                f"{cin}->{cout}@{D}^3 B={B}")

    def fam_conv_wgrad(xt, ldx, dy, lddy, dims, cin, cout, ks, prec):
        B, D, H, W = dims
        v = B * D * H * W
        return (f"conv{ks}x{ks}x{ks}_wgrad", 2.0 * v * cout * cin * ks ** 3, 4.0 * v * (cin + cout) + 4.0 * cin * cout * ks ** 3,
                f"{cin}->{cout}@{D}^3 B={B}")

    pending = []
    saved = {"conv_fwd": wrap("conv_fwd", fam_conv_fwd), "conv_wgrad": wrap("conv_wgrad", fam_conv_wgrad)}
    try:
        for _ in range(reps):
            loss = crit(model(x), y)
            loss.backward()
            model.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
    finally:
        for k, v in saved.items():
            setattr(Fn, k, v)
    per_label = collections.defaultdict(list)
    for fam, e0, e1, flops, nbytes, label in pending:
        per_label[(fam, label)].append((e0.elapsed_time(e1), flops, nbytes))
    # dominant = the (family, shape) with the largest total time
    tot = {k: sum(t for t, _, _ in v) for k, v in per_label.items()}
    (fam, label), _ = max(tot.items(), key=lambda kv: kv[1])
    rows = per_label[(fam, label)]
    avg_ms = sum(t for t, _, _ in rows) / len(rows)
    flops = rows[0][1]
    achieved = flops / (avg_ms * 1e-3) / 1e12
    peak = MFMA_PEAK_TFLOPS[precision] if "wgrad" not in fam or precision == "fp32" else MFMA_PEAK_TFLOPS[precision]
    return {"kernel": f"{fam} {label}", "bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 5), "traffic": None, "avg_ms_per_launch": round(avg_ms, 4),
            "launches_per_step": len(rows) // reps, "algorithmic_flops_per_launch": flops,
            "share_of_timed_kernels": round(tot[(fam, label)] / max(sum(tot.values()), 1e-9), 4)}
