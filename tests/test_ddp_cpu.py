"""world_size-2 gloo test of the bucketed gradient all-reducer (the N>1 path of bench.py): averaged
gradients, unused parameters (cls_token / frozen encoder) and bucket ordering."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        import importlib
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ddp = importlib.import_module("3dmedicalimagesegmentation_amd").ddp
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.randn(s)) for s in [(7, 5), (33,), (4, 4, 3), (1, 1, 9), (129,)]]
        red = ddp.GradAllReducer(params, bucket_bytes=256)
        assert len(red.buckets) >= 2
        assert red.buckets[0].params[0] is params[-1]          # reverse order: last parameter first
        # step 1: every parameter but #3 (the "cls_token") gets a rank-dependent gradient
        loss = sum((p * (rank + 1) * (i + 1)).sum() for i, p in enumerate(params) if i != 3)
        loss.backward()
        red.finish()
        for i, p in enumerate(params):
            if i == 3:
                assert p.grad is None
            else:
                exp = torch.full_like(p, (i + 1) * (1 + world) / 2.0)   # mean over ranks of (rank+1)*(i+1)
                assert torch.allclose(p.grad, exp), (i, p.grad.flatten()[:3], exp.flatten()[:3])
        # step 2: "frozen encoder" -- only the last two parameters receive gradients
        for p in params:
            p.grad = None
        loss = (params[3] * (rank + 2)).sum() + (params[4] * 3).sum()
        loss.backward()
        red.finish()
        assert params[0].grad is None and params[1].grad is None and params[2].grad is None
        assert torch.allclose(params[3].grad, torch.full_like(params[3], (2 + 3) / 2.0))
        assert torch.allclose(params[4].grad, torch.full_like(params[4], 3.0))
        red.broadcast_parameters(params)
        red.remove()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def test_grad_allreducer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def _flat_worker(rank, world, port, q):
    """Arena mode of GradAllReducer on CPU tensors: buckets are slices of one flat gradient buffer, readiness comes
    from functional._ret (immediate) and functional.flush_deferred (deferred weight gradients)."""
    try:
        sys.path.insert(0, ROOT)
        import importlib
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
        Fn, ddp = pkg.functional, pkg.ddp
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.randn(s)) for s in [(6, 4), (10,), (3, 3, 3), (1, 1, 5), (40,)]]
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        fg = torch.zeros(n)
        views = [(p, fg[o:o + p.numel()].view_as(p)) for p, o in zip(params, offs)]
        Fn.register_grad_sinks(views)
        flat = dict(param=torch.zeros(n), grad=fg, offsets=offs, params=params, total=n)
        red = ddp.GradAllReducer(params, bucket_bytes=64, flat=flat)
        assert len(red.buckets) >= 2 and red.buckets[0].params[-1] is params[-1]
        # "backward": write gradients into the arena views in reverse order; #3 never gets one; #0 is deferred
        for i in (4, 2, 1):
            v = Fn._gout(params[i])
            assert v is not None
            v.fill_(float((rank + 1) * (i + 1)))
            assert Fn._ret(params[i], v) is None and params[i].grad is v
        v0 = Fn._gout(params[0])
        assert Fn._ret(params[0], v0, deferred=True) is None
        v0.fill_(float(rank + 1))              # the deferred kernel "runs" here
        Fn._DEFAULT_STATE.defer["armed"] = True
        Fn.flush_deferred()
        red.finish()
        for i in (4, 2, 1, 0):
            exp = (i + 1) * (1 + world) / 2.0
            assert torch.allclose(params[i].grad, torch.full_like(params[i], exp)), (i, params[i].grad.flatten()[:2])
        assert params[3].grad is None
        red.remove()
        Fn.clear_grad_sinks()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def test_grad_allreducer_flat_arena_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flat_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def _plan_worker(rank, world, port, q):
    """bench.py's N>1 communication plan on CPU: AdamW.plan_reduced cuts the arena into ranges at parameter boundaries
    (parameters without gradient excluded), and all-reducing the ranges one by one equals one all-reduce of the buffer."""
    try:
        sys.path.insert(0, ROOT)
        import importlib
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
        torch.manual_seed(0)
        shapes = [(6, 4), (10,), (1, 1, 5), (3, 3, 3), (40,), (7, 9), (128,)]
        params = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + 7) // 8 * 8
        flat = dict(param=torch.zeros(n), grad=torch.zeros(n), offsets=offs, params=params, total=n, shadow=None)
        opt = pkg.AdamW(params, lr=1e-3, flat=flat)
        for i, p in enumerate(params):
            if i != 2:                                   # "cls_token": never gets a gradient
                p.grad = flat["grad"][offs[i]:offs[i] + p.numel()].view_as(p)
        plan = opt.plan_reduced(max_elems=64)
        runs = plan["runs"]
        assert plan["pattern"] == tuple(i != 2 for i in range(len(params)))
        covered = []
        for (i, j, lo, hi) in runs:
            assert lo == offs[i] and hi == offs[j] + (params[j].numel() + 3) // 4 * 4 and lo % 8 == 0
            assert hi - lo <= 64 or i == j               # a single parameter may exceed the cap, several together may not
            covered += list(range(i, j + 1))
        assert covered == [i for i in range(len(params)) if i != 2]         # every gradient exactly once, in order
        buf = torch.arange(n, dtype=torch.float32) * (rank + 1)
        ref = buf.clone()
        dist.all_reduce(ref)
        for (_, _, lo, hi) in runs:
            dist.all_reduce(buf[lo:hi])
        for (_, _, lo, hi) in runs:
            assert torch.equal(buf[lo:hi], ref[lo:hi])
        lo2, hi2 = offs[2], offs[2] + 8
        assert torch.equal(buf[lo2:hi2], torch.arange(n, dtype=torch.float32)[lo2:hi2] * (rank + 1))   # untouched slot
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def test_reduced_plan_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


@pytest.mark.parametrize("config", ["c2", "c5"])
def test_bench_launcher_starts_n_ranks_gloo_stub(config):
    """`python bench.py --gpus 2` without a rendezvous environment must start 2 ranks itself (a child torchrun job) and
    print ONE JSON line from rank 0 -- exercised here on the CPU stub step over gloo (the schedule of the data-parallel
    step: ranges all-reduced as they fill, update in completion order)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--steps", "3", "--warmup", "1", "--config", config],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_world"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["ranks_agree"] is True and out["scaling"] == "weak" and out["value"] > 0
    for k in ("metric", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config"):
        assert k in out
    assert ("c5" in out["config"]["workload"]) == (config == "c5")      # config 5: six passes per step, the frozen ones on one piece


def test_stage_ranges_and_piece_plan():
    """UNETR.stage_ranges tile the arena in backward-completion order and TrainStep's piece plan cuts AdamW runs at piece
    boundaries (host logic only: arenas are faked on CPU tensors)."""
    import importlib
    pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
    ts = pkg.train_step
    cfg = dict(in_channels=1, out_channels=2, img_size=(32, 32, 32), feature_size=16, hidden_size=128, mlp_dim=512,
               num_heads=4, pos_embed="perceptron", norm_name="instance", res_block=True)
    m = pkg.UNETR(**cfg)
    params = list(m.parameters())
    offs, n = [], 0
    for p in params:
        offs.append(n)
        n += (p.numel() + 7) // 8 * 8
    m._flat = dict(param=torch.zeros(n), grad=torch.zeros(n), offsets=offs, params=params, total=n, shadow=None)
    r = m.stage_ranges()
    assert len(r) == 5 and r[0][1] == n and r[4][0] == 0 and [a[0] for a in r[:4]] == [b[1] for b in r[1:]]      # contiguous, reverse order
    names = [k for k, _ in m.named_parameters()]
    first_conv = offs[names.index("encoder1.layer.conv1.conv.weight")]
    assert r[0][0] == first_conv and r[1][0] == offs[names.index("vit.blocks.8.mlp.linear1.weight")]
    assert r[2][0] == offs[names.index("vit.blocks.4.mlp.linear1.weight")]
    assert r[3][0] == offs[names.index("vit.blocks.1.mlp.linear1.weight")]        # the last pass is block 0 + patch embedding only
    tail = ts.split_range(m._flat, *r[4], 2)
    assert tail[0][0] == 0 and tail[-1][1] == r[4][1] and all(a[1] == b[0] for a, b in zip(tail, tail[1:])) and len(tail) == 2
    assert all(lo in offs for lo, _ in tail)
    opt = pkg.AdamW(params, lr=1e-3, flat=m._flat)
    for i, p in enumerate(params):
        if names[i] != "vit.patch_embedding.cls_token":
            p.grad = m._flat["grad"][offs[i]:offs[i] + p.numel()].view_as(p)
    cuts = sorted({lo for lo, _ in r} | {lo for lo, _ in tail})
    plan = opt.plan_reduced(cuts=cuts)
    for (_, _, lo, hi) in plan["runs"]:
        assert not any(lo < c < hi for c in cuts)                     # no run straddles a piece boundary
    covered = sum(hi - lo for _, _, lo, hi in plan["runs"])
    assert covered >= sum(p.numel() for p in params) - 128
