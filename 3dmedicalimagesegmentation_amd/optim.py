"""AdamW with torch.optim.AdamW semantics (unetr_segmentation_3d.py:522: lr from the CLI, weight_decay=1e-5,
betas (0.9, 0.999), eps 1e-8, amsgrad off) running on the HIP kernel ``unetr_adamw`` (csrc/norm_misc.hip).

Per-parameter step counters (a parameter whose grad is None does not step, as in torch) live in one device
tensor and are advanced by a single masked add, so ``step()`` never synchronises with the host and can be
captured in a hipGraph together with forward and backward."""
import torch

from ._capi import call


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._steps = {}   # group index -> flat device tensor of per-parameter step counts
        self._masks = {}   # (group index, pattern) -> 0/1 increment tensor

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        stream = torch.cuda.current_stream().cuda_stream
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            params = group["params"]
            pattern = tuple(p.grad is not None for p in params)
            if not any(pattern):
                continue
            dev = next(p for p in params if p.grad is not None).device
            steps = self._steps.get(gi)
            if steps is None:
                steps = torch.zeros(len(params), dtype=torch.float32, device=dev)
                self._steps[gi] = steps
            mask = self._masks.get((gi, pattern))
            if mask is None:
                mask = torch.tensor([1.0 if f else 0.0 for f in pattern], dtype=torch.float32, device=dev)
                self._masks[(gi, pattern)] = mask
            steps += mask
            for i, p in enumerate(params):
                g = p.grad
                if g is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32:
                    raise RuntimeError("HIP AdamW needs fp32 parameters on a ROCm device (no CPU fallback)")
                if not p.is_contiguous():
                    raise RuntimeError("HIP AdamW needs contiguous parameters")
                if not g.is_contiguous():
                    g = g.contiguous()
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                call("unetr_adamw", p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                     group["lr"], b1, b2, group["eps"], group["weight_decay"], steps.data_ptr() + 4 * i, stream)
        return loss
