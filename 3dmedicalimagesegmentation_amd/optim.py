"""AdamW with torch.optim.AdamW semantics (unetr_segmentation_3d.py:522: lr from the CLI, weight_decay=1e-5,
betas (0.9, 0.999), eps 1e-8, amsgrad off) running on the HIP kernel ``unetr_adamw`` (csrc/norm_misc.hip).

Per-parameter step counters (a parameter whose grad is None does not step, as in torch) live in one device
tensor and are advanced by a single masked add, so ``step()`` never synchronises with the host and can be
captured in a hipGraph together with forward and backward.

``flat=model.use_flat_buffers()``: parameters, gradients and both moments live in flat arenas, and one kernel
launch covers every contiguous run of parameters that received a gradient (2 launches for the full model:
MONAI's unused ``cls_token`` splits the arena), instead of one launch per tensor."""
import torch

from . import functional as Fn
from ._capi import call


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, flat=None):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._steps = {}   # group index -> flat device tensor of per-parameter step counts
        self._masks = {}   # (group index, pattern) -> 0/1 increment tensor
        self._flat = flat
        self._flat_state = None
        if flat is not None:
            if len(self.param_groups) != 1 or [id(p) for p in self.param_groups[0]["params"]] != [id(p) for p in flat["params"]]:
                raise ValueError("flat= needs a single param group holding model.parameters() in order")
            self._host_steps = [0] * len(flat["params"])

    def _advance_steps(self, gi, params, pattern, dev):
        steps = self._steps.get(gi)
        if steps is None:
            steps = torch.zeros(len(params), dtype=torch.float32, device=dev)
            self._steps[gi] = steps
        mask = self._masks.get((gi, pattern))
        if mask is None:
            mask = torch.tensor([1.0 if f else 0.0 for f in pattern], dtype=torch.float32, device=dev)
            self._masks[(gi, pattern)] = mask
        call("unetr_counter_add", steps.data_ptr(), mask.data_ptr(), steps.numel(), torch.cuda.current_stream().cuda_stream)
        return steps

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        stream = torch.cuda.current_stream().cuda_stream
        per_tensor = False
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            params = group["params"]
            pattern = tuple(p.grad is not None for p in params)
            if not any(pattern):
                continue
            dev = next(p for p in params if p.grad is not None).device
            if self._flat is not None and self._flat_step(group, params, pattern, dev, stream):
                continue
            per_tensor = True
            steps = self._advance_steps(gi, params, pattern, dev)
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
                     group["lr"], b1, b2, group["eps"], group["weight_decay"], steps.data_ptr() + 4 * i,
                     Fn.shadow_ptr_for_update(p), stream)
        if self._flat is not None and per_tensor:
            Fn.refresh_x3_shadow(self._flat)      # (per-tensor launches do not write the bf16x3 word shadow: one derive launch)
        Fn.refresh_conv_packs()          # one grouped launch: packed conv weights follow the update
        return loss

    def _flat_runs(self, params, pattern, max_elems=None, cuts=()):
        """Contiguous arena ranges (i, j, lo, hi) of parameters that step together: consecutive parameters with a
        gradient and equal step counts, optionally cut at parameter boundaries into pieces of <= max_elems and at the
        arena offsets listed in ``cuts`` (the boundaries of the data-parallel communication pieces)."""
        offs = self._flat["offsets"]
        cuts = set(cuts)
        runs, i, n = [], 0, len(params)
        while i < n:
            if not pattern[i]:
                i += 1
                continue
            j = i
            while (j + 1 < n and pattern[j + 1] and self._host_steps[j + 1] == self._host_steps[i]
                   and offs[j + 1] not in cuts and (max_elems is None or offs[j + 1] + params[j + 1].numel() - offs[i] <= max_elems)):
                j += 1
            runs.append((i, j, offs[i], offs[j] + (params[j].numel() + 3) // 4 * 4))
            i = j + 1
        return runs

    def _launch_run(self, group, run, steps, gptr, g_bf16, gscale, stream):
        flat = self._flat
        i, j, lo, hi = run
        m, v = self._flat_state
        b1, b2 = group["betas"]
        shadow = flat.get("shadow")            # bf16 copy of the arena read by the bf16-storage GEMMs, refreshed in the same kernel
        sptr = shadow.data_ptr() + lo * 2 if shadow is not None else None
        words = flat.get("shadow_x3")          # bf16x3 mode: the word shadow (functional.weight_x3), written by the same kernel
        wptr = words.data_ptr() + lo * 4 if words is not None else None
        call("unetr_adamw_reduced", flat["param"].data_ptr() + lo * 4, gptr + lo * (2 if g_bf16 else 4), int(g_bf16), gscale,
             m.data_ptr() + lo * 4, v.data_ptr() + lo * 4, hi - lo, group["lr"], b1, b2, group["eps"], group["weight_decay"],
             steps.data_ptr() + 4 * i, sptr, wptr, stream)
        for k in range(i, j + 1):
            self._host_steps[k] += 1

    def _flat_step(self, group, params, pattern, dev, stream):
        """One launch per contiguous run of parameters with arena gradients and equal step counts."""
        flat = self._flat
        gbase = flat["grad"].data_ptr()
        for p, o, has in zip(params, flat["offsets"], pattern):
            if has and p.grad.data_ptr() != gbase + o * 4:
                return False          # some gradient is not in the arena (e.g. accumulated): per-tensor path
        if self._flat_state is None:
            self._flat_state = (torch.zeros_like(flat["param"]), torch.zeros_like(flat["param"]))
        steps = self._advance_steps(0, params, pattern, dev)
        for run in self._flat_runs(params, pattern):
            self._launch_run(group, run, steps, gbase, False, 1.0, stream)
        if flat.get("shadow") is not None:           # the kernel rewrote the bf16 shadow slices of every parameter that stepped
            Fn.mark_flat_maintained([p for p, has in zip(params, pattern) if has], flat.get("state"))
        Fn.refresh_x3_shadow(flat, written=True)     # (bf16x3 mode: the optimizer kernels wrote the word shadow)
        return True

    # ---- single-GPU fused step: AdamW of the ViT Linear weights rides on the grouped weight-gradient launch ----------------
    # begin_fused_step(pattern) AFTER the forward and BEFORE backward (step counters advance, the arena state is handed to
    # functional.flush_deferred; the model's next forward drops an arming whose step never finished),
    # finish_fused_step() AFTER backward: one table-driven launch updates every parameter the fused launch did not cover.
    # The gradients of the fused weights are never written: p.grad of those parameters names a stale arena slice.
    @torch.no_grad()
    def begin_fused_step(self, pattern):
        from . import _capi
        flat = self._flat
        if flat is None or len(self.param_groups) != 1 or flat.get("state") is None:
            raise RuntimeError("begin_fused_step needs AdamW(..., flat=model.use_flat_buffers())")
        group = self.param_groups[0]
        params = group["params"]
        pattern = tuple(bool(f) for f in pattern)
        # everything that can be checked BEFORE backward is checked here: the epilogue updates the ViT weights during backward,
        # so a failure found afterwards (finish_fused_step) leaves the model half-updated
        for p, has in zip(params, pattern):
            if has and p.grad is not None:
                raise RuntimeError("fused step: a planned parameter already carries .grad (its gradient would be accumulated outside "
                                   "the arena); call zero_grad(set_to_none=True) before the step")
            if self.state.get(p):
                raise RuntimeError("fused step: this optimizer has per-tensor AdamW state (an earlier step() took the per-tensor path); "
                                   "the arena moments would silently restart from zero")
        if self._flat_state is None:
            self._flat_state = (torch.zeros_like(flat["param"]), torch.zeros_like(flat["param"]))
        steps = self._advance_steps(0, params, pattern, flat["param"].device)
        m, v = self._flat_state
        shadow = flat.get("shadow")
        b1, b2 = group["betas"]
        words = flat.get("shadow_x3")
        arena = _capi.AdamWArena(flat["param"].data_ptr(), flat["grad"].data_ptr(), m.data_ptr(), v.data_ptr(),
                                 shadow.data_ptr() if shadow is not None else None, steps.data_ptr(), flat["param"].numel(),
                                 group["lr"], b1, b2, group["eps"], group["weight_decay"], words.data_ptr() if words is not None else None)
        gbase = flat["grad"].data_ptr()
        index = {gbase + o * 4: i for i, (o, has) in enumerate(zip(flat["offsets"], pattern)) if has}
        self._fused = dict(pattern=pattern, arena=arena, index=index, done=[], keep=(m, v, steps))
        flat["state"].fuse = self._fused

    @torch.no_grad()
    def finish_fused_step(self):
        """A RuntimeError raised here is fatal for the step: the weight-gradient epilogue has already updated the fused weights
        and the step counters have advanced, the remaining parameters have not stepped -- reload a checkpoint."""
        import ctypes
        flat, fz = self._flat, self._fused
        flat["state"].fuse = None
        self._fused = None
        group = self.param_groups[0]
        params = group["params"]
        pattern = fz["pattern"]
        if tuple(p.grad is not None for p in params) != pattern:
            raise RuntimeError("fused step: the set of parameters that received gradients differs from the planned pattern")
        gbase = flat["grad"].data_ptr()
        done = set(fz["done"])
        self.fused_parameters = len(done)          # how many parameters the weight-gradient launch updated itself in this step
        for i, (p, o, has) in enumerate(zip(params, flat["offsets"], pattern)):
            if has and i not in done and p.grad.data_ptr() != gbase + o * 4:
                raise RuntimeError("fused step: a gradient was produced outside the arena")
        rest = tuple(has and i not in done for i, has in enumerate(pattern))
        # a run shares ONE step counter (its first parameter's): the runs depend on which neighbours have equal step counts, so
        # their boundaries are part of the key (alternating patterns -- the feat / recon passes of the ranking pre-training --
        # make counts diverge between two uses of the same pattern)
        runs = self._flat_runs(params, rest)
        key = (pattern, tuple(sorted(done)), tuple((i, j) for i, j, _, _ in runs))
        cache = getattr(self, "_range_tables", None)
        if cache is None:
            cache = self._range_tables = {}
        ent = cache.get(key)
        if ent is None:
            rows, blocks = [], 0
            for i, j, lo, hi in runs:
                rows.append((lo, hi, i, blocks))
                blocks += (hi - lo + 4095) // 4096
            table = torch.tensor(rows, dtype=torch.int64, device=flat["param"].device) if rows else None
            ent = cache[key] = (table, len(rows), blocks)
        table, nr, blocks = ent
        if nr:
            call("unetr_adamw_ranges", ctypes.byref(fz["arena"]), table.data_ptr(), nr, blocks, torch.cuda.current_stream().cuda_stream)
        for k, has in enumerate(pattern):
            if has:
                self._host_steps[k] += 1
        Fn.refresh_x3_shadow(flat, written=True)
        if flat.get("shadow") is not None:
            Fn.mark_flat_maintained([p for p, has in zip(params, pattern) if has], flat.get("state"))
        Fn.refresh_conv_packs()

    # ---- data-parallel arena update: all-reduce pieces overlap the optimizer kernels of the pieces before them ----
    def plan_reduced(self, max_elems=None, cuts=(), pattern=None):
        """Freeze a gradient pattern (default: which parameters have .grad now) into arena ranges of <= max_elems that never
        straddle an offset in ``cuts``.  The plan is reused every step by ``step_reduced`` / ``step_runs`` -- under hipGraph
        replay ``.grad`` attributes do not change."""
        if self._flat is None or len(self.param_groups) != 1:
            raise RuntimeError("plan_reduced needs AdamW(..., flat=model.use_flat_buffers())")
        params = self.param_groups[0]["params"]
        pattern = tuple(p.grad is not None for p in params) if pattern is None else tuple(bool(f) for f in pattern)
        return dict(pattern=pattern, runs=self._flat_runs(params, pattern, max_elems, cuts))

    # ---- the reduced step in pieces (train_step.TrainStep, data-parallel form): begin -> step_runs per piece -> end, each on
    # whatever stream is current (the communication stream, right behind the piece's all-reduce)
    @torch.no_grad()
    def begin_reduced_step(self, plan):
        if self._flat_state is None:
            self._flat_state = (torch.zeros_like(self._flat["param"]), torch.zeros_like(self._flat["param"]))
        return self._advance_steps(0, self.param_groups[0]["params"], plan["pattern"], self._flat["param"].device)

    @torch.no_grad()
    def step_runs(self, plan, ks, steps, gsrc, gscale):
        g_bf16 = gsrc.dtype == torch.bfloat16
        if not g_bf16 and gsrc.dtype != torch.float32:
            raise RuntimeError("gradient source must be fp32 or bf16")
        stream = torch.cuda.current_stream().cuda_stream
        for k in ks:
            self._launch_run(self.param_groups[0], plan["runs"][k], steps, gsrc.data_ptr(), g_bf16, gscale, stream)

    @torch.no_grad()
    def end_reduced_step(self, plan):
        Fn.refresh_x3_shadow(self._flat, written=True)
        if self._flat.get("shadow") is not None:
            Fn.mark_flat_maintained([p for p, has in zip(self.param_groups[0]["params"], plan["pattern"]) if has], self._flat.get("state"))
        Fn.refresh_conv_packs()

    @torch.no_grad()
    def step_reduced(self, plan, gsrc, gscale, before_run=None, order=None):
        """AdamW over the planned ranges with gradients read from ``gsrc`` (a flat fp32 or bf16 tensor laid out like
        the arena: the all-reduced communication buffer) times ``gscale``.  ``before_run(k, lo, hi)`` runs before range
        k's kernel is launched -- the caller makes the current stream wait for that range's all-reduce there."""
        group = self.param_groups[0]
        params = group["params"]
        dev = self._flat["param"].device
        stream = torch.cuda.current_stream().cuda_stream
        if self._flat_state is None:
            self._flat_state = (torch.zeros_like(self._flat["param"]), torch.zeros_like(self._flat["param"]))
        steps = self._advance_steps(0, params, plan["pattern"], dev)
        g_bf16 = gsrc.dtype == torch.bfloat16
        if not g_bf16 and gsrc.dtype != torch.float32:
            raise RuntimeError("gradient source must be fp32 or bf16")
        runs = plan["runs"]
        for k in (order if order is not None else range(len(runs))):      # order: the order the pieces finish reducing in
            run = runs[k]
            if before_run is not None:
                before_run(k, run[2], run[3])
            self._launch_run(group, run, steps, gsrc.data_ptr(), g_bf16, gscale, stream)
        Fn.refresh_x3_shadow(self._flat, written=True)
        if self._flat.get("shadow") is not None:
            Fn.mark_flat_maintained([p for p, has in zip(params, plan["pattern"]) if has], self._flat.get("state"))
        Fn.refresh_conv_packs()
