"""bench.py -- UNETR training step throughput on MI355X (the metric of BASELINE.json).

A "step" = forward + DiceCE loss + backward + AdamW on one batch of synthetic 96^3 CT-like volumes
(config[1]: UNETR(img 96, patch 16, hidden 768, 12 layers, 12 heads, 4 classes), batch 2 per GPU), with
inputs resident in HBM.  N GPUs = N ranks (one per GPU, RCCL), weak scaling: each rank keeps batch 2 and
gradients are all-reduced (bucketed, side stream).  Prints ONE JSON line on rank 0.

Extra objects: "roofline" for the dominant kernel (timed live with HIP events on the launch stream) and
"cpu_baseline" (the CPU oracle = the reference's operator graph in plain PyTorch, timed on the host cores
on a bounded sample; rank 0, N=1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = dict(in_channels=1, out_channels=4, img_size=(96, 96, 96), feature_size=16, hidden_size=768, mlp_dim=3072,
           num_heads=12, pos_embed="perceptron", norm_name="instance", res_block=True)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2, help="volumes per GPU")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a captured hipGraph")
    ap.add_argument("--no-flat", action="store_true", help="per-tensor grads/AdamW instead of the flat arenas")
    ap.add_argument("--force-dist", action="store_true", help="run the N>1 code path (two graphs + eager all-reduce slot) on one rank")
    ap.add_argument("--fp32-comm", action="store_true", help="all-reduce gradients in fp32 even in bf16 mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    return ap.parse_args()


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    # RCCL / HIP print banners on stdout; the contract is ONE JSON line there, so keep the real stdout aside
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist_on = world > 1 or (args.force_dist and "RANK" in os.environ)   # --force-dist under torchrun: 1-rank RCCL group
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
    from oracle.unetr_oracle import synthetic_volume  # data generator only (shared with the tests)

    torch.manual_seed(1234)  # same initial weights on every rank
    model = pkg.UNETRLogits(**CFG).to(dev)
    model.precision = args.precision
    crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)
    flat = None if args.no_flat else model.use_flat_buffers()
    opt = pkg.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, flat=flat)
    x, y = synthetic_volume(args.batch, 1, 96, 4, seed=1234 + rank)
    x, y = x.to(dev), y.to(dev)

    # ---- data parallel (N > 1): weak scaling, batch `--batch` per rank, gradients averaged over ranks ----------
    # Arena mode: hipGraph = forward + loss + backward (+ one cast of the gradient arena into the bf16 communication
    # buffer); then the buffer is all-reduced in <= 6 pieces on a side stream (eager RCCL calls) while AdamW runs on each
    # piece as soon as it is reduced, reading the summed gradients straight from the buffer.  No collective is captured
    # inside a hipGraph, so the path does not depend on RCCL's capture support.
    # --force-dist exercises exactly this path with a single rank (what the 1-GPU box can test).
    ddp_on = dist_on or args.force_dist
    comm_dtype = torch.bfloat16 if (args.precision == "bf16" and not args.fp32_comm) else torch.float32
    reducer = None
    comm_buf = None
    if ddp_on and flat is None:
        if not dist_on:
            raise SystemExit("--force-dist needs the arena mode (drop --no-flat)")
        reducer = pkg.ddp.GradAllReducer(model.parameters(), process_group=None)
    if ddp_on and flat is not None:
        comm_buf = torch.zeros(flat["total"], dtype=comm_dtype, device=dev)
        if dist_on:
            for p in model.parameters():
                dist.broadcast(p.data, src=0)
            pkg.functional.invalidate_weight_shadows()

    comm_stream = torch.cuda.Stream() if ddp_on else None
    plan = {"p": None}
    in_place = comm_buf is not None and comm_dtype == torch.float32     # fp32 comm: all-reduce the gradient arena itself

    def fwd_bwd():
        logit_map = model(x)
        loss = crit(logit_map, y)
        loss.backward()
        if comm_buf is not None and not in_place:
            comm_buf.copy_(flat["grad"])        # fp32 -> bf16 communication buffer
        return loss

    def comm_update():
        """Arena data-parallel step: the gradient buffer is all-reduced in <= 6 pieces on a side stream while the main
        stream runs AdamW on each piece as soon as its all-reduce has finished (the optimizer kernel reads the summed
        gradients from the communication buffer and averages on the fly: no copy back, no separate scaling pass)."""
        buf = flat["grad"] if in_place else comm_buf
        if plan["p"] is None:
            plan["p"] = opt.plan_reduced(max_elems=(flat["total"] + 5) // 6)
        runs = plan["p"]["runs"]
        main = torch.cuda.current_stream()
        comm_stream.wait_stream(main)
        events = []
        with torch.cuda.stream(comm_stream):
            for (_, _, lo, hi) in runs:
                if dist_on:
                    dist.all_reduce(buf[lo:hi], op=dist.ReduceOp.SUM)
                ev = torch.cuda.Event()
                ev.record(comm_stream)
                events.append(ev)
        opt.step_reduced(plan["p"], buf, 1.0 / world, before_run=lambda k, lo, hi: main.wait_event(events[k]))

    def comm():
        if reducer is not None:
            reducer.finish()

    def update():
        opt.step()
        opt.zero_grad(set_to_none=True)

    def step():
        loss = fwd_bwd()
        if comm_buf is not None:
            comm_update()
            if graph is None:                     # eager mode re-plans every step from fresh .grad attributes
                plan["p"] = None
                opt.zero_grad(set_to_none=True)
        else:
            comm()
            update()
        return loss

    graph = None
    use_graph = not args.no_graph and reducer is None
    # eager warm-up (allocates workspaces / optimizer state / RCCL communicators; needed before graph capture)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            loss = step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    log(f"eager warm-up done, loss {float(loss.item()):.5f}")
    if use_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            if comm_buf is None:
                with torch.cuda.graph(graph):
                    loss = step()
            else:
                with torch.cuda.graph(graph):
                    loss = fwd_bwd()
                plan["p"] = None                  # planned from the .grad attributes the captured backward just set
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                print(f"[bench] graph capture failed ({type(e).__name__}: {e}); falling back to eager", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    def run_step():
        if graph is None:
            step()
        elif comm_buf is None:
            graph.replay()
        else:
            graph.replay()
            comm_update()

    log("graph captured" if graph is not None else "eager mode")
    for _ in range(args.warmup):
        run_step()

    def sync_all():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    sync_all()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    log(f"timed region done: {dt / args.steps * 1e3:.3f} ms/step")
    ms_per_step = dt / args.steps * 1e3
    value = world * args.batch * args.steps / dt

    out = {
        "metric": "training volumes/sec (96^3, 4-class)", "value": round(value, 3), "unit": "volumes/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": "UNETR(img=96^3,patch=16,hidden=768,layers=12,heads=12,classes=4) fwd+DiceCE+bwd+AdamW, "
                               f"batch {args.batch}/GPU, configs[1]", "global_batch": world * args.batch,
                   "launch": ("hipGraph" if comm_buf is None else "hipGraph(fwd+bwd) + chunked all-reduce overlapped with AdamW") if graph is not None else "eager",
                   "grad_comm_dtype": (str(comm_dtype).replace("torch.", "") if ddp_on else None), "final_loss": float(loss.item())},
    }

    if rank == 0 and not args.no_roofline:
        try:
            from tools.roofline import dominant_kernel_roofline
            out["roofline"] = dominant_kernel_roofline(pkg, model, crit, x, y, args.precision)
        except Exception as e:  # noqa: BLE001
            out["roofline"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            from tools.roofline import encoder_forward_rate
            out["encoder_fwd"] = encoder_forward_rate(pkg, model, x, args.precision)
            # the same encoder kernels at a token count where the GEMMs are throughput- rather than latency-bound
            xb = x[:1].expand(32, -1, -1, -1, -1).contiguous()
            out["encoder_fwd_batch32"] = encoder_forward_rate(pkg, model, xb, args.precision, iters=5)
            del xb
        except Exception as e:  # noqa: BLE001
            out["encoder_fwd"] = {"error": f"{type(e).__name__}: {e}"}
    log("roofline done")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
        log("cpu baseline done")

    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


def cpu_baseline(args):
    """The CPU oracle (plain PyTorch fp32, the reference's operator graph) on this host's cores."""
    from oracle.unetr_oracle import OracleUNETR, oracle_train_step, synthetic_volume
    # the GPU box gives one GPU a 16-CPU share; os.cpu_count() reports the whole host and oversubscribes badly
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    torch.manual_seed(1234)
    ref = OracleUNETR(**CFG)
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-4, weight_decay=1e-5)
    x, y = synthetic_volume(1, 1, 96, 4, seed=1234)
    oracle_train_step(ref, opt, x, y)  # warm-up
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        oracle_train_step(ref, opt, x, y)
    dt = time.perf_counter() - t0
    return {"value": round(args.cpu_steps / dt, 4), "unit": "volumes/s", "cores": cores, "kind": "port",
            "sample": f"{args.cpu_steps} fwd+DiceCE+bwd+AdamW steps of the same 96^3 UNETR at batch 1 (1 warm-up), torch fp32, "
                      f"{cores} threads"}


if __name__ == "__main__":
    main()
