"""The training step of the reference loop (unetr_segmentation_3d.py:220-226: forward -> DiceCE -> backward ->
AdamW step -> zero_grad) as ONE object that owns how it is launched on an MI355X:

* single GPU: the whole step -- forward, loss, backward, AdamW -- is captured into one hipGraph and replayed (no per-step
  host work, no host sync: AdamW's step counters live on the device, the loss stays a device scalar).  ``fuse_update``: AdamW
  of the ViT weight matrices is applied in the epilogue of the grouped weight-gradient launch that ends backward (their
  gradients are never stored); one table-driven launch updates the remaining parameters.  Same bits as the separate launch.
* data parallel (one process per GPU, RCCL over xGMI; new capability, the reference is single-GPU): backward runs in the
  five passes of ``UNETR.forward_staged`` and the sum-all-reduce of each pass's gradient range (``UNETR.stage_ranges``)
  is issued on a side HIP stream as soon as that pass has FINISHED -- the launching thread runs one graph ahead and waits for
  the event behind the pass before (``handover="host"``: no stream waits on an unfinished event of another stream) -- so it
  runs underneath the passes that follow (conv-side gradients under ViT blocks 11..8, those under blocks 7..4, ...).  What
  is left when backward ends -- the last range (block 0 + patch embedding, 41 MB of the 370 MB), cut into ``tail_pieces`` --
  overlaps with the AdamW kernels of the ranges already reduced: the optimizer kernel reads the summed gradients straight
  from the communication buffer and averages on the fly, so there is no copy back and no scaling pass.  Every pass that
  hands a range over ends a hipGraph (pass 0 shares one with pass 1: 4 graphs, one memory pool); the collectives are
  ordinary eager RCCL calls between graph launches, nothing depends on capturing a collective.  Gradients travel in fp32
  by default (the same sum the single-GPU arithmetic would do), bf16 on request -- then the weight-gradient launch writes the
  bf16 gradients of the ViT weights straight into the communication buffer (``fuse_comm``).

``bench.py`` and ``tests/test_model_gpu.py`` both drive the step through this class, so what is measured is what is tested.
"""
import torch

from . import functional as Fn


_SIDE = {}


def side_stream(device=None):
    """ONE side stream per device for every TrainStep of the process: eager warm-up and graph capture both run on it; it shares the
    device's scratch workspace with the default stream (functional.share_workspace), which is allocated during the eager warm-up
    -- i.e. OUTSIDE any graph's private memory pool -- and used by all steps and graphs."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    st = _SIDE.get(idx)
    if st is None:
        st = _SIDE[idx] = torch.cuda.Stream(device=idx)
        Fn.share_workspace(st)          # ordered against the launching stream by construction: one scratch buffer for both
    return st


def split_range(flat, lo, hi, pieces):
    """cut arena range [lo, hi) at parameter boundaries into <= pieces parts of roughly equal size"""
    offs = [o for o in flat["offsets"] if lo < o < hi]
    cuts, target = [lo], (hi - lo) / float(pieces)
    for o in offs:
        if len(cuts) < pieces and o - lo >= target * len(cuts) and o > cuts[-1]:
            cuts.append(o)
    cuts.append(hi)
    return [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]


class TrainStep:
    def __init__(self, model, criterion, optimizer, x, y, *, use_graph=True, data_parallel=False, process_group=None,
                 comm_dtype=torch.float32, tail_pieces=2, warmup=2, overlap_update=False, fuse_update=False, fuse_comm=True,
                 handover="host", loss_fn=None, freeze_encoder=False, extra_cuts=()):
        """loss_fn(enc4, logits) -> scalar loss: replaces ``criterion(logits, y)`` (the ranking pre-training step,
        unetr_ranking_pretraining_3d.py:259-262, takes its loss from enc4 in the "feat" stage and from the logits in the "recon"
        stage); the model must then return the (enc4, logits) tuple.  freeze_encoder: the forward of this step runs
        ``model(x, freeze_encoder=True)`` (:262): only the decoder side receives gradients -- in the data-parallel form backward is then
        ONE pass whose communication pieces are the arena runs that received a gradient.  extra_cuts: arena offsets at which AdamW
        launches must be split besides the communication-piece boundaries -- steps with different gradient patterns that share an
        optimizer (feat / recon) advance the step counters of different parameter sets, and a launch applies ONE counter to its run.
        fuse_update (single GPU, flat arenas): AdamW of the ViT Linear weights (92 % of the parameters) is applied in the
        epilogue of the grouped weight-gradient launch that ends backward -- their gradients are never stored or re-read (8 of
        34 bytes per weight) and the optimizer's streaming hides under that launch's MFMA work; one table-driven AdamW launch
        covers the rest.  Same bits as the unfused step.  ``p.grad`` of the fused weights is NOT valid afterwards.
        handover (data parallel, captured form): how the communication stream learns that a backward pass has ended.  "stream" =
        ``comm_stream.wait_stream(main)`` after every graph launch: the cross-stream wait costs the MAIN stream ~55 us each on
        this stack.  "host" (default) = an event behind every graph; the launching thread runs one graph ahead, waits for the
        event of the pass before -- the GPU is busy with the next graph meanwhile -- and only then issues that pass's all-reduce
        and AdamW launches, so no stream ever waits on an unfinished event of another (-0.27 ms per step on one GPU).
        fuse_comm (data parallel, bf16 gradient communication): the grouped weight-gradient launch of every backward pass writes
        bf16 gradients straight into the communication buffer (no fp32 gradient store, no cast pass over them); one table-driven
        cast per piece covers the other parameters.  Same bits in the communication buffer as the separate cast.
        overlap_update (single GPU): the data-parallel launch form without a process group -- backward in passes, AdamW on
        each pass's arena range on the side stream underneath the passes that follow -- captured as ONE hipGraph in which the
        side stream is a branch: the bandwidth-bound optimizer kernels hide under the latency-bound ViT backward chain."""
        flat = getattr(model, "_flat", None)
        self.model, self.crit, self.opt, self.x, self.y = model, criterion, optimizer, x, y
        self.loss_fn = loss_fn
        self.frozen = bool(freeze_encoder)
        if self.frozen and loss_fn is None:
            raise ValueError("freeze_encoder needs loss_fn(enc4, logits): the frozen forward returns the tuple")
        if self.frozen and overlap_update:
            raise ValueError("overlap_update is the staged form: not available with freeze_encoder")
        self.flat = flat
        self.fuse = bool(fuse_update) and not data_parallel and not overlap_update
        if self.fuse and flat is None:
            raise RuntimeError("fuse_update needs model.use_flat_buffers() and AdamW(flat=...)")
        self._fuse_pattern = None      # which parameters receive gradients: known after the first (unfused) eager step
        self.dp = bool(data_parallel) or bool(overlap_update)
        self.overlap = bool(overlap_update) and not data_parallel
        self.group = process_group
        self.world = 1
        self.dist = None
        if self.dp:
            if flat is None:
                raise RuntimeError("the data-parallel step needs model.use_flat_buffers() and AdamW(flat=...)")
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.dist = dist
                self.world = dist.get_world_size(process_group)
        if handover not in ("host", "stream"):
            raise ValueError("handover must be 'host' or 'stream'")
        self.handover = handover
        self.comm_dtype = comm_dtype
        self.in_place = comm_dtype == torch.float32
        self.fuse_comm = bool(fuse_comm) and self.dp and comm_dtype == torch.bfloat16 and not self.overlap
        self._comm_fuse = None         # armed for the backward passes of a step once the gradient pattern is known
        self._cast_tables = {}
        self.one_graph = False
        self.loss = None
        self._one = torch.ones((), dtype=torch.float32, device=x.device)
        self.graphs = None
        self.stages = None
        self.eager_steps = 0
        self.first_loss = None
        if self.dp:
            self.comm_stream = torch.cuda.Stream()
            self.comm_buf = None if self.in_place else torch.zeros(flat["total"], dtype=comm_dtype, device=flat["grad"].device)
            if self.frozen:
                # one backward pass; its pieces = the arena runs that receive a gradient, known after the first eager step
                self.pieces, self.npass, self._first_k = [[]], 1, 0
                self.cuts = sorted(set(extra_cuts))
            else:
                ranges = model.stage_ranges()
                # per backward pass.  The conv side's 16 MB wait for pass 1 and travel with its range: one hand-over to the
                # communication stream less per step (each costs the main stream ~70 us), nothing lost in overlap
                self.pieces = [[], [ranges[0], ranges[1]]] + [[r] for r in ranges[2:-1]] + [split_range(flat, *ranges[-1], tail_pieces)]
                self.npass = len(self.pieces)
                self._first_k = next(k for k, st in enumerate(self.pieces) if st)
                self.cuts = sorted({lo for st in self.pieces for lo, _ in st} | {hi for st in self.pieces for _, hi in st} | set(extra_cuts))
            self._plan = None         # the AdamW launches: planned from the gradient pattern of the first (eager) step
            self._steps = None
        # (gradient attributes another step object left behind -- a captured step never runs its closing zero_grad in Python -- would
        # be read as part of THIS step's gradient pattern)
        optimizer.zero_grad(set_to_none=True)
        # eager warm-up: allocates workspaces, optimizer state, RCCL communicators -- all of which must exist before capture
        self.side = side = side_stream(x.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            Fn.workspace(x.device)                      # exists before any capture on this stream
            # (fused update: the first eager step runs unfused and records the gradient pattern, the second builds the range
            # table of the fused form -- both must exist before capture)
            for _ in range(max(2 if (self.fuse or self.fuse_comm) else 1, warmup)):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if use_graph:
            self._capture()
            if self.dp:
                optimizer.zero_grad(set_to_none=True)      # the captured passes attached arena views as .grad; replays never read them

    # ------------------------------------------------------------------------------------------- single GPU
    def _loss_of(self, enc4, logits):
        return self.loss_fn(enc4, logits) if self.loss_fn is not None else self.crit(logits, self.y)

    def _fwd_loss(self):
        out = self.model(self.x, freeze_encoder=True) if self.frozen else self.model(self.x)
        if self.loss_fn is not None and not isinstance(out, tuple):
            raise RuntimeError("loss_fn(enc4, logits) needs a model that returns the (enc4, logits) tuple (UNETR, not UNETRLogits)")
        return self._loss_of(*out) if isinstance(out, tuple) else self.crit(out, self.y)

    def _backward(self):
        # (the seed gradient is a persistent device scalar: loss.backward() alone launches a fill for it every step)
        if self._one is None or self._one.device != self.loss.device:
            self._one = torch.ones((), dtype=self.loss.dtype, device=self.loss.device)
        self.loss.backward(self._one)

    def _single_step(self):
        if self.fuse and self._fuse_pattern is not None:
            self.loss = self._fwd_loss()
            self.opt.begin_fused_step(self._fuse_pattern)      # (armed after the forward: UNETR.forward drops a stale arming)
            self._backward()
            self.opt.finish_fused_step()
        else:
            self.loss = self._fwd_loss()
            self._backward()
            if self.fuse:
                self._fuse_pattern = tuple(p.grad is not None for p in self.opt.param_groups[0]["params"])
            self.opt.step()
        self.opt.zero_grad(set_to_none=True)

    # ---------------------------------------------------------------------------------------- data parallel
    def _pass0(self):
        if self.frozen:
            enc4, logits = self.model(self.x, freeze_encoder=True)
            self.stages = []
        else:
            enc4, logits, self.stages = self.model.forward_staged(self.x)
        if self.fuse_comm and self._plan is not None:
            if self._comm_fuse is None:
                flat = self.flat
                gbase = flat["grad"].data_ptr()
                index = {gbase + o * 4: i for i, (o, has) in enumerate(zip(flat["offsets"], self._plan["pattern"])) if has}
                self._comm_fuse = dict(kind="bf16out", grad=gbase, out=self.comm_buf.data_ptr(), total=flat["param"].numel(), index=index, done=[])
            self._comm_fuse["done"].clear()               # what THIS step's weight-gradient launches write into the communication buffer
            self.flat["state"].fuse = self._comm_fuse
        self.loss = self._loss_of(enc4, logits)
        self._backward()
        if self.npass == 1 and self._comm_fuse is not None and self.flat["state"].fuse is self._comm_fuse:
            self.flat["state"].fuse = None             # (single-pass form: there is no later pass to disarm the epilogue)

    def _pass(self, k):
        st = self.stages[k - 1]
        roots = [r for r, leaf in st if leaf.grad is not None]
        if roots:
            torch.autograd.backward(roots, [leaf.grad for r, leaf in st if leaf.grad is not None])
        if k == self.npass - 1 and self._comm_fuse is not None and self.flat["state"].fuse is self._comm_fuse:
            self.flat["state"].fuse = None             # the last pass of the step has queued its weight-gradient launch
            # the cast tables skip exactly the ranges the epilogue wrote when they were built: a step whose fused set differs (a
            # problem dropping off the bf16 grouped path) would send stale bf16 values through the all-reduce
            done = frozenset(self._comm_fuse["done"])
            for (lo, hi), ent in self._cast_tables.items():
                if ent[3] != frozenset(i for i in done if lo <= self.flat["offsets"][i] < hi):
                    raise RuntimeError("data-parallel step: the set of gradients written by the weight-gradient epilogue changed between steps")

    def _reduce_and_update(self, k, after=None):
        """Pass k has been launched on the current stream (after: the event recorded behind it -- the host waits for it here and
        the communication stream is not made to wait on the main stream; None: stream-side wait).  On the communication stream, behind it: all-reduce each of its
        gradient pieces and run AdamW on the piece right after its all-reduce -- stream order is the only synchronisation
        (one cross-stream wait per pass; an event per piece plus a wait per AdamW launch cost 0.36 ms per step), and the
        optimizer work of passes 0-2 runs underneath the backward passes that follow."""
        if not self.pieces[k]:
            return
        main = torch.cuda.current_stream()
        if after is None:
            self.comm_stream.wait_stream(main)
        else:
            after.synchronize()
        g = self.flat["grad"]
        src = g if self.in_place else self.comm_buf
        with torch.cuda.stream(self.comm_stream):
            if k == self._first_k:
                self._steps = self.opt.begin_reduced_step(self._plan)
            # every piece's collective is ISSUED first (async_op=True: c10d enqueues it on the process group's own stream
            # behind an event recorded on the current = communication stream, and hands back a Work), so the collectives of
            # one pass run back to back; then, piece by piece, Work.wait() makes the communication stream wait for that
            # collective's end event (ProcessGroupNCCL::WorkNCCL::synchronizeStream: a stream wait, no host block) and AdamW
            # on the piece follows in stream order -- underneath the next piece's collective.
            works = []
            for lo, hi in self.pieces[k]:
                buf = src[lo:hi]
                if not self.in_place:
                    if self._comm_fuse is not None:      # (armed in every step since the pattern is known: eager, captured, replayed)
                        self._cast_rest(lo, hi)
                    else:
                        Fn.cast_bf16(g[lo:hi], out=buf)
                works.append(self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
                             if self.dist is not None else None)
            for (lo, hi), work in zip(self.pieces[k], works):
                if work is not None:
                    work.wait()
                self.opt.step_runs(self._plan, self._runs_of[(lo, hi)], self._steps, src, 1.0 / self.world)
            if k == len(self.pieces) - 1:
                self.opt.end_reduced_step(self._plan)
        if k == len(self.pieces) - 1:
            main.wait_stream(self.comm_stream)        # the next forward reads the updated parameters

    def _cast_rest(self, lo, hi):
        """fp32 -> bf16 of the gradient ranges of piece [lo, hi) that the weight-gradient epilogue did not already write into the
        communication buffer (everything but the ViT weight matrices), as ONE table-driven launch"""
        ent = self._cast_tables.get((lo, hi))
        if ent is None:
            flat = self.flat
            params = flat["params"]
            done = sorted(flat["offsets"][i] for i in set(self._comm_fuse["done"]) if lo <= flat["offsets"][i] < hi)
            numel = {flat["offsets"][i]: (params[i].numel() + 7) // 8 * 8 for i in set(self._comm_fuse["done"])}
            rows, blocks, cur = [], 0, lo
            for o in done + [hi]:
                if o > cur:
                    rows.append((cur, o, blocks))
                    blocks += (o - cur + 8191) // 8192
                cur = max(cur, o + numel.get(o, 0))
            table = torch.tensor(rows, dtype=torch.int64, device=flat["param"].device) if rows else None
            ent = self._cast_tables[(lo, hi)] = (table, len(rows), blocks,
                                                 frozenset(i for i in set(self._comm_fuse["done"]) if lo <= flat["offsets"][i] < hi))
        table, nr, blocks = ent[:3]
        if nr:
            Fn.call("unetr_cast_bf16_ranges", self.flat["grad"].data_ptr(), self.comm_buf.data_ptr(), table.data_ptr(), nr, blocks,
                    torch.cuda.current_stream().cuda_stream)

    def _make_plan(self):
        """which parameters step (MONAI's ViT carries an unused cls_token: no gradient, skipped like torch.optim.AdamW does),
        as AdamW launches that never straddle a communication piece"""
        self._plan = self.opt.plan_reduced(cuts=self.cuts)
        runs = self._plan["runs"]
        if self.frozen:
            self.pieces = [[(r[2], r[3]) for r in runs]]
        self._runs_of = {(lo, hi): [k for k, r in enumerate(runs) if lo <= r[2] and r[3] <= hi] for st in self.pieces for lo, hi in st}
        covered = sorted(k for ks in self._runs_of.values() for k in ks)
        assert covered == list(range(len(runs))), "every AdamW run must lie inside exactly one communication piece"

    def _dp_step_eager(self):
        first = self._plan is None
        self._pass0()
        if not first:
            self._reduce_and_update(0)
        for k in range(1, self.npass):
            self._pass(k)
            if not first:
                self._reduce_and_update(k)
        if first:                         # the pattern is only known once every pass has run: this one step is not overlapped
            self._make_plan()
            for k in range(self.npass):
                self._reduce_and_update(k)
        elif tuple(p.grad is not None for p in self.opt.param_groups[0]["params"]) != self._plan["pattern"]:
            now = [p.grad is not None for p in self.opt.param_groups[0]["params"]]
            diff = [(i, bool(a), bool(b)) for i, (a, b) in enumerate(zip(now, self._plan["pattern"])) if bool(a) != bool(b)]
            raise RuntimeError("data-parallel step: the set of parameters that receive gradients changed between steps "
                               f"(parameter index, has a gradient now, had one when the step was planned): {diff[:12]}")
        self.opt.zero_grad(set_to_none=True)

    def _eager_step(self):
        self.eager_steps += 1
        if self.dp:
            self._dp_step_eager()
        else:
            self._single_step()
        if self.eager_steps == 1:
            self.first_loss = self.loss.detach().clone()      # loss at the initial weights (bench.py: vs the CPU oracle's)

    # ------------------------------------------------------------------------------------------------ graphs
    def _capture(self):
        # thread-local capture mode: with a process group alive, its watchdog thread polls the events of the warm-up collectives
        # (hipEventQuery) -- under the default global mode such a call from another thread during capture is an error and aborts
        # the process (seen once in six 1-rank RCCL runs)
        mode = "thread_local"
        if not self.dp:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.side, capture_error_mode=mode):
                self._single_step()
            self.graphs = [g]
            return
        if self.overlap and self.dist is None:
            # no collective between the passes: the whole step is one graph, the communication stream a branch of it (forked by
            # comm_stream.wait_stream(main) after every pass, joined by main.wait_stream(comm_stream) after the last)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.side, capture_error_mode=mode):
                self._pass0()
                self._reduce_and_update(0)
                for k in range(1, self.npass):
                    self._pass(k)
                    self._reduce_and_update(k)
            self.graphs = [g]
            self.one_graph = True
            return
        # one graph per hand-over to the communication stream: passes that hand nothing over (pass 0: the conv side's range
        # travels with pass 1) are captured together with the pass that follows them -- every extra graph in the chain costs
        # the main stream ~25 us, every hand-over ~55 us (tools/probe_dp_split.py)
        groups, cur = [], []
        for k in range(self.npass):
            cur.append(k)
            if self.pieces[k] or k == self.npass - 1:
                groups.append(cur)
                cur = []
        graphs, pool = [], None
        for ks in groups:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, stream=self.side, capture_error_mode=mode):
                for k in ks:
                    if k == 0:
                        self._pass0()
                    else:
                        self._pass(k)
            pool = g.pool()
            graphs.append(g)
        self.graphs, self.graph_passes = graphs, groups

    def run(self):
        """one training step; returns nothing (self.loss is the device scalar of this step)"""
        if self.graphs is None:
            self._eager_step()
        elif not self.dp or self.one_graph:
            self.graphs[0].replay()
        else:
            if self.handover == "stream":
                for g, ks in zip(self.graphs, self.graph_passes):
                    g.replay()
                    self._reduce_and_update(ks[-1])
                return
            main, pending = torch.cuda.current_stream(), None
            for g, ks in zip(self.graphs, self.graph_passes):
                g.replay()
                ev = torch.cuda.Event()
                ev.record(main)
                if pending is not None:                 # the pass before this one: its event is waited for while this graph runs
                    self._reduce_and_update(*pending)
                pending = (ks[-1], ev)
            self._reduce_and_update(*pending)

    @property
    def launch(self):
        if self.graphs is None:
            return "eager"
        if not self.dp:
            fused = self.fuse and getattr(self.opt, "fused_parameters", 0) > 0     # (fp32 mode has no bf16-storage weight-gradient launch to ride on)
            return "hipGraph(fwd+loss+bwd+AdamW" + (", AdamW of the ViT weights in the weight-gradient epilogue)" if fused else ")")
        if self.one_graph:
            return (f"hipGraph(fwd+loss+bwd in {self.npass} passes; AdamW per pass on a side-stream branch underneath the passes that follow)")
        if self.frozen:
            return "1 hipGraph (fwd+loss+bwd, encoder frozen), all-reduce of the decoder's gradient runs on a side stream, AdamW per reduced piece"
        return (f"{len(self.graphs)} hipGraphs ({self.npass} backward passes: fwd+loss+conv side | ViT passes 1-{self.npass - 1}), "
                "per-pass all-reduce on a side stream, AdamW per reduced piece")
