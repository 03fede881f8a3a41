"""bench.py -- UNETR training step throughput on MI355X (the metric of BASELINE.json).

A "step" = forward + DiceCE loss + backward + AdamW on one batch of synthetic 96^3 CT-like volumes
(config[1]: UNETR(img 96, patch 16, hidden 768, 12 layers, 12 heads, 4 classes), batch 2 per GPU), with
inputs resident in HBM.  N GPUs = N ranks (one process per GPU, RCCL), weak scaling: each rank keeps batch 2 and
gradients are all-reduced per backward pass on a side stream, underneath the rest of backward (train_step.py).

    python bench.py --gpus N --steps K --warmup W

* started WITHOUT a rendezvous environment (no RANK) and N > 1: this process touches no GPU; it starts
  ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same args>`` as a
  child, rank 0 of which prints the JSON line, and exits with the child's code.
* started BY torchrun (RANK / LOCAL_RANK / WORLD_SIZE set): one rank of the job.

Prints ONE JSON line on rank 0.  Extra objects: "roofline" for the dominant kernel of the step (tools/roofline.py),
"families" (ms/step per kernel family), "encoder_fwd*" (ViT encoder forward alone vs the bf16 MFMA peak) and
"cpu_baseline" (the CPU oracle = the reference's operator graph in plain PyTorch, timed on the host cores on a bounded
sample; rank 0, N=1 only -- the only place bench.py touches oracle/).

--config c2 (default) | c4 (160^3, encoder activation checkpointing, batch 1) | c5 (ranking pre-training step at 96^3:
[4,1,96^3] batch, 'feat' then 'recon' stage of unetr_ranking_pretraining_3d.py:238-296, all three slice axes).
--stub: a tiny CPU step over gloo that exercises launcher + bucket schedule + JSON contract without a GPU (tests).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = dict(in_channels=1, out_channels=4, img_size=(96, 96, 96), feature_size=16, hidden_size=768, mlp_dim=3072,
           num_heads=12, pos_embed="perceptron", norm_name="instance", res_block=True)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="volumes per GPU (default 2; c4: 1; c5: 4)")
    ap.add_argument("--config", default="c2", choices=["c2", "c4", "c5"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "bf16x3"])
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of captured hipGraphs")
    ap.add_argument("--no-flat", action="store_true", help="per-tensor grads/AdamW instead of the flat arenas (N=1 only)")
    ap.add_argument("--force-dist", action="store_true", help="run the N>1 code path (staged backward + per-pass all-reduce slots) on one rank")
    ap.add_argument("--overlap-update", type=int, default=0, help="N=1: AdamW per backward pass on a side-stream branch of the ONE captured graph "
                    "(measured SLOWER on MI355X / ROCm 7.2: 5.46 vs 5.00 ms per step -- a branch in a hipGraph costs more than the optimizer kernels it hides)")
    ap.add_argument("--fuse-update", type=int, default=1, help="N=1: AdamW of the ViT Linear weights in the epilogue of the grouped weight-gradient "
                    "launch (same bits as the separate optimizer launch; 0 = separate)")
    ap.add_argument("--handover", default="host", choices=["host", "stream"], help="N>1: how the communication stream learns that a backward pass "
                    "has ended (host: the launching thread waits for the pass's event one graph behind; stream: cross-stream wait per pass)")
    ap.add_argument("--fuse-comm", type=int, default=1, help="N>1 with bf16 gradient communication: the weight-gradient epilogue writes bf16 gradients "
                    "straight into the communication buffer (same bits as the separate cast; 0 = fp32 store + cast pass)")
    ap.add_argument("--bf16-comm", action="store_true", help="all-reduce gradients in bf16 (the default when more than one rank runs in bf16 mode)")
    ap.add_argument("--fp32-comm", action="store_true", help="all-reduce gradients in fp32 (the single-GPU arithmetic; twice the xGMI bytes: "
                    "370 MB per step, more than the backward passes it has to hide under -- DESIGN.md section 7)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the bf16x3 (tolerance-grade) leg reported as parity_mode")
    ap.add_argument("--cpu-steps", type=int, default=10, help="timed CPU-oracle steps (after 3 warm-up steps; SURVEY 8d)")
    ap.add_argument("--cpu-warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=5, help="timed windows of exactly --steps steps each; the reported ms_per_step / value are "
                    "the MEDIAN window's (min / max beside it): one 0.1 s window moves by more than most single optimisations")
    ap.add_argument("--stub", action="store_true", help="CPU/gloo stub step (launcher + schedule test, no GPU)")
    return ap.parse_args(argv)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args):
    """--gpus N > 1 without a rendezvous environment: start N ranks as a CHILD torchrun job.  Nothing in this process has
    touched (or will touch) the GPU, and it never replaces itself with another program."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log(f"launcher: starting {args.gpus} ranks: {' '.join(cmd)}")
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------------ stub rank
def stub_rank(args, rank, world, real_stdout):
    """The schedule of the data-parallel step on CPU tensors over gloo: 'backward' fills the gradient arena range by range
    (in the order of UNETR.stage_ranges), each range is all-reduced asynchronously as soon as it is full, the update
    consumes the ranges in completion order.  Checks: every rank ends with identical parameters."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1 << 16
    bounds = [0, n // 8, n // 2, 3 * n // 4, n]
    ranges = [(bounds[i], bounds[i + 1]) for i in (3, 2, 1, 0)]
    # --config c5: the six passes of the ranking pre-training step -- per slice axis a "feat" pass (every range but the decoder's
    # share of the conv side, in backward order) and a "recon" pass (encoder frozen: the decoder's share only, ONE piece)
    dec = (bounds[3] + (n - bounds[3]) // 2, n)
    passes = [ranges] if args.config != "c5" else [[(bounds[3], dec[0])] + ranges[1:], [dec]] * 3
    torch.manual_seed(0)
    param = torch.randn(n)
    grad = torch.zeros(n)
    t0 = None
    for it in range(args.warmup + args.steps):
        if it == args.warmup:
            dist.barrier()
            t0 = time.perf_counter()
        for prs in passes:
            works = []
            for lo, hi in prs:
                grad[lo:hi] = torch.sin(param[lo:hi]) * (rank + 1)           # this rank's "backward pass" over the range
                works.append((lo, hi, dist.all_reduce(grad[lo:hi], async_op=True)))
            for lo, hi, w in works:
                w.wait()
                param[lo:hi] -= 1e-2 * grad[lo:hi] / world
    dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    cs = torch.tensor([param.double().sum().item()], dtype=torch.float64)
    lo_, hi_ = cs.clone(), cs.clone()
    dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
    out = {"metric": "stub steps/sec (CPU, gloo)", "value": round(world * args.steps / t.item(), 3), "unit": "steps/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(t.item() / args.steps * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": ("stub: 4 gradient ranges, async all-reduce per range, update in completion order" if args.config != "c5" else
                                   "stub c5: 3 x (feat pass: 4 ranges in backward order + recon pass: 1 decoder range), async all-reduce per range")},
           "rccl_world": dist.get_world_size(), "backend": "gloo", "ranks_agree": bool(lo_.item() == hi_.item())}
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    return 0


# ------------------------------------------------------------------------------------------------ one rank
def main():
    args = parse()
    rendezvous = "RANK" in os.environ
    if args.gpus > 1 and not rendezvous:
        sys.exit(launch_ranks(args))

    # RCCL / HIP print banners on stdout; the contract is ONE JSON line there, so keep the real stdout aside
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.stub:
        sys.exit(stub_rank(args, rank, world, real_stdout))

    import torch
    dist_on = world > 1 or (args.force_dist and rendezvous)   # --force-dist under torchrun: a real 1-rank RCCL group
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if dist_on:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: RCCL world size {dist.get_world_size()} != --gpus {args.gpus}")

    pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
    from tools.synthetic import synthetic_volume

    ddp_on = dist_on or args.force_dist
    if args.config == "c5":
        out = run_c5(args, pkg, dev, rank, world, dist if dist_on else None)
    else:
        out = run_supervised(args, pkg, dev, rank, world, dist, ddp_on, synthetic_volume)
    out["rccl_world"] = dist.get_world_size() if dist_on else 1
    if out["rccl_world"] != out["n_gpus"]:
        raise SystemExit(f"bench.py: the JSON line would report n_gpus {out['n_gpus']} over an RCCL world of {out['rccl_world']}")
    names = [torch.cuda.get_device_name(dev)]
    if dist_on:
        gathered = [None] * world
        dist.all_gather_object(gathered, f"rank {rank}: cuda:{local_rank} {names[0]}")
        names = gathered
    out["devices"] = names

    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


def run_supervised(args, pkg, dev, rank, world, dist, ddp_on, synthetic_volume):
    import torch
    cfg = dict(CFG)
    batch = args.batch if args.batch is not None else (1 if args.config == "c4" else 2)
    if args.config == "c4":
        cfg["img_size"] = (160, 160, 160)
    size = cfg["img_size"][0]
    torch.manual_seed(1234)  # same initial weights on every rank
    model = pkg.UNETRLogits(**cfg).to(dev)
    model.precision = args.precision
    model.encoder_checkpointing = args.config == "c4"
    crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)
    if args.no_flat and ddp_on:
        raise SystemExit("the data-parallel step needs the flat arenas (drop --no-flat)")
    flat = None if args.no_flat else model.use_flat_buffers()
    opt = pkg.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, flat=flat)
    x, y = synthetic_volume(batch, 1, size, 4, seed=1234 + rank)
    x, y = x.to(dev), y.to(dev)
    if dist is not None:
        for p in model.parameters():
            dist.broadcast(p.data, src=0)
        pkg.functional.invalidate_weight_shadows()      # p.data was rewritten behind the version counters
        if flat is not None:
            flat["shadow"].copy_(flat["param"])         # (and the arena's bf16 shadow follows)

    # gradient communication type: bf16 by default for real multi-rank runs in bf16 mode (185 MB per step instead of 370 MB: the
    # fp32 form cannot hide under backward); fp32 on request and for the 1-rank --force-dist form
    bf16_comm = args.precision == "bf16" and not args.fp32_comm and (args.bf16_comm or world > 1)
    comm_dtype = torch.bfloat16 if bf16_comm else torch.float32
    graph_err = None
    try:
        step = pkg.TrainStep(model, crit, opt, x, y, use_graph=not args.no_graph, data_parallel=ddp_on, comm_dtype=comm_dtype,
                             overlap_update=bool(args.overlap_update) and flat is not None and not ddp_on,
                             fuse_update=bool(args.fuse_update) and flat is not None and not ddp_on and not args.overlap_update,
                             fuse_comm=bool(args.fuse_comm), handover=args.handover)
    except Exception as e:  # noqa: BLE001
        if args.no_graph:
            raise
        graph_err = f"{type(e).__name__}: {e}"
        log(f"graph capture failed ({graph_err}); falling back to eager launches")
        torch.cuda.synchronize()
        if flat is not None:
            flat["state"].reset_deferred()              # a failed capture may have left deferred weight-gradient work queued
        opt.zero_grad(set_to_none=True)
        step = pkg.TrainStep(model, crit, opt, x, y, use_graph=False, data_parallel=ddp_on, comm_dtype=comm_dtype)
    # (after a failed capture the weights have already stepped: this step's first loss is NOT the loss at the initial weights)
    first_loss = float(step.first_loss.item()) if graph_err is None else None
    log(f"{step.launch}; first-step loss {first_loss}")
    for _ in range(args.warmup):
        step.run()

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # each window: EXACTLY --steps steps between barrier + device synchronisation on both sides, MAX over ranks
    wins = []
    for _ in range(max(1, args.windows)):
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step.run()
        sync_all()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        wins.append(dt)
    dt = sorted(wins)[len(wins) // 2]
    ms_per_step = dt / args.steps * 1e3
    log(f"timed region done: {ms_per_step:.3f} ms/step (median of {len(wins)} windows of {args.steps} steps: "
        f"{min(wins) / args.steps * 1e3:.3f} .. {max(wins) / args.steps * 1e3:.3f})")
    value = world * batch * args.steps / dt
    tag = {"c2": "configs[1]", "c4": "configs[3]: 160^3, encoder activation checkpointing"}[args.config]
    out = {
        "metric": f"training volumes/sec ({size}^3, 4-class)", "value": round(value, 3), "unit": "volumes/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "windows": len(wins), "ms_per_step_min": round(min(wins) / args.steps * 1e3, 4), "ms_per_step_max": round(max(wins) / args.steps * 1e3, 4),
        "config": {"workload": f"UNETR(img={size}^3,patch=16,hidden=768,layers=12,heads=12,classes=4) fwd+DiceCE+bwd+AdamW, "
                               f"batch {batch}/GPU, {tag}", "global_batch": world * batch, "launch": step.launch,
                   "grad_comm_dtype": (str(comm_dtype).replace("torch.", "") if ddp_on else None),
                   "first_step_loss": first_loss, "final_loss": float(step.loss.item()), "graph_capture_error": graph_err},
    }
    if rank == 0 and not args.no_roofline and args.config == "c2" and not ddp_on:
        try:
            from tools import roofline as rl
            out.update(rl.step_report(pkg, step, batch, args.precision, ms_per_step))
        except Exception as e:  # noqa: BLE001
            out["roofline"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            from tools import roofline as rl
            out.update(rl.encoder_report(pkg, model, x, args.precision))
        except Exception as e:  # noqa: BLE001
            out["encoder_fwd"] = {"error": f"{type(e).__name__}: {e}"}
        log("roofline done")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, cfg, batch, first_loss)
        log("cpu baseline done")
    if rank == 0 and world == 1 and args.config == "c2" and args.precision == "bf16" and not args.no_parity_mode and not args.no_roofline:
        try:
            out["parity_mode"] = parity_mode(args, pkg, dev, cfg, batch, synthetic_volume)
        except Exception as e:  # noqa: BLE001
            out["parity_mode"] = {"error": f"{type(e).__name__}: {e}"}
        log("parity mode done")
    return out


def run_c5(args, pkg, dev, rank, world, dist=None):
    """BASELINE config[4]: one pre-training step of unetr_ranking_pretraining_3d.py:238-296 at 96^3 = for each of the
    three slice axes (:241) a 'feat' update (loss on enc4, :259-260) and a 'recon' update (loss on the logits with the
    encoder frozen, :261-262) on a [4, 1, 96^3] batch (2 volumes x 2 transforms) -- six forward/backward/AdamW passes."""
    import torch
    from tools.synthetic import synthetic_volume
    torch.manual_seed(1234)
    cfg = dict(CFG, out_channels=2)                     # Task09 spleen: n_classes = 2 (:303)
    model = pkg.UNETR(**cfg).to(dev)
    model.precision = args.precision
    flat = None if args.no_flat else model.use_flat_buffers()      # arena gradients: AdamW per contiguous run, grouped ViT weight gradients
    opt = pkg.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, flat=flat)
    x, _ = synthetic_volume(4, 1, 96, 2, seed=1234 + rank)
    x = x.to(dev)
    if dist is not None:
        if flat is None:
            raise SystemExit("the data-parallel pre-training step needs the flat arenas (drop --no-flat)")
        for p in model.parameters():
            dist.broadcast(p.data, src=0)
        pkg.functional.invalidate_weight_shadows()
        flat["shadow"].copy_(flat["param"])

    patterns = {}          # stage -> which parameters receive gradients (recorded by the first, unfused, step)

    def step():
        last = None
        for axis in (2, 3, 4):
            for stage in ("feat", "recon"):
                fused = dist is None and stage in patterns
                if stage == "feat":
                    inp, _ = model(x)
                    init_idx = 1                           # enc4 is 12^3: partition size 3
                else:
                    _, inp = model(x, freeze_encoder=True)
                    init_idx = 7                           # logits are 96^3: partition size 24
                loss = pkg.ranking_loss(inp, axis, init_idx, 0.1, kind="ranking")
                if fused:
                    opt.begin_fused_step(patterns[stage])      # (after the forward, which drops any stale arming)
                if fused:
                    # (the optimizer is armed: AdamW of the ViT weights rides on the weight-gradient launch that
                    # ends this backward pass -- the three "feat" passes; the frozen-encoder passes update the conv side only)
                    loss.backward()
                    opt.finish_fused_step()
                else:
                    loss.backward()
                    if flat is not None and args.fuse_update:
                        patterns[stage] = tuple(p.grad is not None for p in opt.param_groups[0]["params"])
                    opt.step()
                opt.zero_grad(set_to_none=True)
                last = loss
        return last

    launch, graph_err = "eager launches", None
    if dist is not None:
        # data parallel: every one of the six passes is a TrainStep in its data-parallel launch form -- "feat": backward in the five
        # passes of UNETR.forward_staged, each pass's gradient range all-reduced (async) on the communication stream underneath the
        # passes that follow, AdamW per reduced piece; "recon" (encoder frozen: decoder gradients only): one backward pass, its
        # gradient runs all-reduced and stepped piece by piece.  Volumes are independent: the sum is averaged in the optimizer kernel.
        bf16_comm = args.precision == "bf16" and not args.fp32_comm and (args.bf16_comm or world > 1)
        comm_dtype = torch.bfloat16 if bf16_comm else torch.float32
        stages = []
        for axis in (2, 3, 4):
            for stage in ("feat", "recon"):
                idx = 1 if stage == "feat" else 7          # enc4 is 12^3: partition size 3; logits are 96^3: partition size 24
                fn = (lambda e, l, a=axis, i=idx: pkg.ranking_loss(e, a, i, 0.1, kind="ranking")) if stage == "feat" else \
                     (lambda e, l, a=axis, i=idx: pkg.ranking_loss(l, a, i, 0.1, kind="ranking"))
                stages.append(pkg.TrainStep(model, None, opt, x, None, use_graph=not args.no_graph, data_parallel=True, comm_dtype=comm_dtype,
                                            fuse_comm=bool(args.fuse_comm), handover=args.handover, loss_fn=fn,
                                            freeze_encoder=stage == "recon", warmup=2))
        launch = f"6 data-parallel TrainSteps ({stages[0].launch} | recon: {stages[1].launch}), gradient communication in {comm_dtype}".replace("torch.", "")

        def run():
            for st in stages:
                st.run()
            return stages[-1].loss
    else:
        for _ in range(2):
            loss = step()
        torch.cuda.synchronize()
        run = step
    if not args.no_graph and flat is not None and dist is None:      # (collectives are eager calls between graph launches: see TrainStep)
        # the six passes (each forward + loss + backward + AdamW, with its own gradient pattern) as ONE hipGraph
        try:
            side = pkg.train_step.side_stream(dev)
            side.wait_stream(torch.cuda.current_stream())
            g = torch.cuda.CUDAGraph()
            holder = {}
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                holder["loss"] = step()
            launch = "hipGraph(6 x (fwd+loss+bwd+AdamW))"

            def run():
                g.replay()
                return holder["loss"]
        except Exception as e:  # noqa: BLE001
            graph_err = f"{type(e).__name__}: {e}"
            log(f"c5 graph capture failed ({graph_err}); eager launches")
            torch.cuda.synchronize()
            flat["state"].reset_deferred()
            opt.zero_grad(set_to_none=True)
    for _ in range(max(1, args.warmup)):
        loss = run()

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # as run_supervised: windows of EXACTLY --steps steps between barrier + device synchronisation, MAX over ranks, median window
    wins = []
    for _ in range(max(1, args.windows)):
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = run()
        sync_all()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        wins.append(dt)
    dt = sorted(wins)[len(wins) // 2]
    return {"metric": "ranking pre-training volumes/sec (96^3)", "value": round(world * 4 * 6 * args.steps / dt, 3), "unit": "volumes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "windows": len(wins), "ms_per_step_min": round(min(wins) / args.steps * 1e3, 4), "ms_per_step_max": round(max(wins) / args.steps * 1e3, 4),
            "config": {"workload": "configs[4]: ranking pre-training, [4,1,96^3] batch, 3 slice axes x (feat + recon) = 6 "
                                   f"fwd/BT-loss/bwd/AdamW passes per step, {launch}",
                       "final_loss": float(loss.item()),
                       "graph_capture_error": graph_err}}


_CPU_REF = {}


def parity_mode(args, pkg, dev, cfg, batch, synthetic_volume):
    """The tolerance-grade mode next to the bf16 headline: the same step in bf16x3 mode (fp32 storage, operands split into bf16
    hi/lo pairs inside the kernels, fp32 accumulation) -- its ms/step, and the relative error of the logits at the initial
    weights (seed 1234, same volumes) against the CPU oracle's, for this mode and for the benched bf16 mode (north_star: 1e-3)."""
    import torch
    x, y = synthetic_volume(batch, 1, cfg["img_size"][0], 4, seed=1234)
    x, y = x.to(dev), y.to(dev)
    ref = _CPU_REF.get("logits0")
    errs = {}
    for mode in ("bf16x3", args.precision):
        torch.manual_seed(1234)
        m = pkg.UNETRLogits(**cfg).to(dev)
        m.precision = mode
        with torch.no_grad():
            lg = m(x)
        if ref is not None:
            errs[mode] = float(((lg.float().cpu() - ref).abs().max() / ref.abs().max()).item())
        del m, lg
    torch.manual_seed(1234)
    model = pkg.UNETRLogits(**cfg).to(dev)
    model.precision = "bf16x3"
    flat = model.use_flat_buffers()
    opt = pkg.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, flat=flat)
    step = pkg.TrainStep(model, pkg.DiceCELoss(to_onehot_y=True, softmax=True), opt, x, y, use_graph=not args.no_graph, fuse_update=True)
    first = float(step.first_loss.item())
    for _ in range(3):
        step.run()
    torch.cuda.synchronize()
    wins = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            step.run()
        torch.cuda.synchronize()
        wins.append((time.perf_counter() - t0) / 10 * 1e3)
    pkg.functional.clear_grad_sinks(flat["state"])
    return {"mode": "bf16x3", "dtype": "fp32 storage, bf16 (hi, lo) split operands, fp32 accumulate", "ms_per_step": round(sorted(wins)[1], 4),
            "volumes_per_s": round(batch / (sorted(wins)[1] * 1e-3), 2), "first_step_loss": first, "tolerance": 1e-3,
            "logits_rel_err_vs_cpu_oracle": errs.get("bf16x3"),
            "bench_mode_logits_rel_err_vs_cpu_oracle": errs.get(args.precision),
            "note": "logits at the initial weights, max |diff| / max |ref| against the CPU oracle (null without the cpu_baseline leg)"}


def cpu_baseline(args, cfg, batch, gpu_first_loss):
    """The CPU oracle (plain PyTorch fp32, the reference's operator graph) on this host's cores: 1 warm-up + --cpu-steps
    timed steps at the bench batch.  Its first-step loss (same seed-1234 weights and data as the GPU run) is reported next
    to the GPU's: a whole-model parity figure that costs nothing inside the timed region."""
    import torch
    from oracle.unetr_oracle import OracleUNETR, oracle_train_step, synthetic_volume
    # the GPU box gives one GPU a 16-CPU share; os.cpu_count() reports the whole host and oversubscribes badly
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    torch.manual_seed(1234)
    ref = OracleUNETR(**cfg)
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-4, weight_decay=1e-5)
    size = cfg["img_size"][0]
    if size > 96:
        batch, steps, warm = 1, 1, 1
    else:
        steps, warm = args.cpu_steps, max(1, args.cpu_warmup)
    x, y = synthetic_volume(batch, 1, size, 4, seed=1234)
    if size <= 96:
        with torch.no_grad():                       # logits at the initial weights: what parity_mode() holds the GPU modes to
            _CPU_REF["logits0"] = ref(x)[1].clone()
    l0 = float(oracle_train_step(ref, opt, x, y))  # first warm-up step; also the first-step loss
    t1 = time.perf_counter()
    for _ in range(warm - 1):
        oracle_train_step(ref, opt, x, y)
    per = (time.perf_counter() - t1) / max(1, warm - 1)
    if warm > 1 and per * steps > 60.0:             # keep the default run within a few minutes on a slow host
        steps = max(3, int(60.0 / per))
    t0 = time.perf_counter()
    for _ in range(steps):
        oracle_train_step(ref, opt, x, y)
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 4), "unit": "volumes/s", "cores": cores, "kind": "port",
            "sample": f"{steps} fwd+DiceCE+bwd+AdamW steps of the same {size}^3 UNETR at batch {batch} ({warm} warm-up), torch fp32, "
                      f"{cores} threads",
            "first_step_loss": l0,
            "first_step_loss_rel_diff_vs_gpu": None if gpu_first_loss is None else abs(l0 - gpu_first_loss) / abs(l0)}


if __name__ == "__main__":
    main()
