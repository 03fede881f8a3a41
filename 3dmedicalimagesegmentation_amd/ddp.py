"""Data-parallel gradient exchange for the UNETR training step: one process per GPU, RCCL over xGMI.

The path shards naturally (volumes are independent: InstanceNorm/LayerNorm are per-sample and DiceCE is a
mean over (b, c)), so the only collective is a sum all-reduce of the gradients, averaged over ranks.  It
is issued per bucket on a side HIP stream as soon as backward has produced every gradient of the bucket
(post-accumulate-grad hooks), so communication overlaps the rest of backward; the optimiser waits on the
side stream in ``finish()``.

Buckets are built in reverse parameter order (gradients become ready decoder-first, then ViT blocks
11 -> 0, patch embedding last).  Parameters that receive no gradient in a step (MONAI's unused
``cls_token``; the whole encoder under ``freeze_encoder=True``) contribute zeros and keep ``grad is None``,
so every rank issues identical collectives.  The reference has no distributed code;
this is new capability required by BASELINE.json's north_star.
"""
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("params", "offsets", "flat", "pending", "launched")

    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        self.offsets = []
        n = 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4  # keep every slice 16-byte aligned
        self.flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        self.pending = len(params)
        self.launched = False


class _FlatBucket:
    """A contiguous slice of the model's flat gradient arena (UNETR.use_flat_buffers): reduced in place."""
    __slots__ = ("params", "flat", "pending", "launched")

    def __init__(self, params, flat_slice):
        self.params = params
        self.flat = flat_slice
        self.pending = len(params)
        self.launched = False


class GradAllReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], process_group: Optional[dist.ProcessGroup] = None,
                 bucket_bytes: int = 32 << 20, flat=None):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("GradAllReducer needs an initialised torch.distributed process group")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        plist = [p for p in params if p.requires_grad]
        if not plist:
            raise ValueError("no trainable parameters")
        self.flat = flat
        if flat is not None:
            self._init_flat(plist, flat, bucket_bytes)
            return
        self.buckets: List[_Bucket] = []
        cur, cur_bytes = [], 0
        for p in reversed(plist):
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self.buckets.append(_Bucket(cur))
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(_Bucket(cur))
        self._where = {}
        for b in self.buckets:
            for p in b.params:
                self._where[p] = b
        self.is_cuda = plist[0].is_cuda
        self.stream = torch.cuda.Stream(device=plist[0].device) if self.is_cuda else None
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in plist]

    def _init_flat(self, plist, flat, bucket_bytes):
        """Zero-copy mode: buckets are contiguous arena slices (built from the END of the parameter list, the order
        in which backward fills them); readiness comes from functional._ret, not from autograd hooks."""
        from . import functional as Fn
        fparams, offs, fg = flat["params"], flat["offsets"], flat["grad"]
        if [id(p) for p in fparams] != [id(p) for p in plist]:
            raise ValueError("flat= needs model.parameters() in order (all trainable)")
        self.buckets = []
        hi = len(fparams) - 1
        while hi >= 0:
            lo, nbytes = hi, fparams[hi].numel() * 4
            while lo > 0 and nbytes < bucket_bytes:
                lo -= 1
                nbytes += fparams[lo].numel() * 4
            end = offs[hi] + (fparams[hi].numel() + 3) // 4 * 4
            self.buckets.append(_FlatBucket(fparams[lo:hi + 1], fg[offs[lo]:end]))
            hi = lo - 1
        self._where = {}
        for b in self.buckets:
            for p in b.params:
                self._where[p] = b
        self.is_cuda = fg.is_cuda
        self.stream = torch.cuda.Stream(device=fg.device) if self.is_cuda else None
        self._hooks = []
        self._cb = self._on_grad
        self._state = flat.get("state") or Fn._DEFAULT_STATE      # the model's ArenaState (UNETR.use_flat_buffers)
        self._state.ready_cb.append(self._cb)

    # broadcast rank 0's parameters so every rank starts from the same weights
    def broadcast_parameters(self, params: Iterable[torch.nn.Parameter]):
        for p in params:
            dist.broadcast(p.data, src=0, group=self.group)
        from . import functional as Fn
        Fn.invalidate_weight_shadows()      # p.data was rewritten behind the version counter

    def _on_grad(self, p):
        b = self._where[p]
        b.pending -= 1
        if b.pending == 0 and not b.launched:
            self._launch(b)

    def _launch(self, b):
        b.launched = True
        if self.flat is not None:
            if self.is_cuda:
                self.stream.wait_stream(torch.cuda.current_stream())
                ctx = torch.cuda.stream(self.stream)
            else:
                ctx = _null()
            with ctx, torch.no_grad():
                work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                work.wait()
                b.flat.div_(self.world)
            return
        have = [p.grad is not None for p in b.params]
        if self.is_cuda:
            self.stream.wait_stream(torch.cuda.current_stream())
            ctx = torch.cuda.stream(self.stream)
        else:
            ctx = _null()
        with ctx, torch.no_grad():
            for p, off, h in zip(b.params, b.offsets, have):
                sl = b.flat[off:off + p.numel()]
                if h:
                    sl.copy_(p.grad.reshape(-1))
                else:
                    sl.zero_()
            work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            work.wait()  # stream-level wait on GPU (no host block); host-level on gloo
            b.flat.div_(self.world)
            for p, off, h in zip(b.params, b.offsets, have):
                if h:
                    p.grad.copy_(b.flat[off:off + p.numel()].view_as(p.grad))

    def finish(self):
        """Call after backward(), before optimizer.step(): flushes buckets that never filled (unused
        parameters) and makes the compute stream wait for all reductions."""
        for b in self.buckets:
            if not b.launched:
                self._launch(b)
        if self.is_cuda:
            torch.cuda.current_stream().wait_stream(self.stream)
        for b in self.buckets:
            b.pending = len(b.params)
            b.launched = False

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self.flat is not None:
            if self._cb in self._state.ready_cb:
                self._state.ready_cb.remove(self._cb)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
