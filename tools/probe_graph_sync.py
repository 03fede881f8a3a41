"""Cost of handing work from a hipGraph on the main stream to a side stream (the data-parallel step does it four times per
step): plain wait_stream after the launch vs an external event recorded INSIDE the captured graph vs nothing."""
import time, torch
dev = torch.device("cuda:0")
a = torch.randn(2048, 2048, device=dev)
outs = [torch.empty_like(a) for _ in range(4)]
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
dummy = torch.zeros(1 << 20, device=dev)


def body(k):
    x = a
    for _ in range(60):
        x = torch.mm(x, a) * 1e-3
    outs[k].copy_(x)


def make(ext):
    gs, evs = [], []
    for k in range(4):
        body(k); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        ev = torch.cuda.Event(external=True) if ext else None
        with torch.cuda.graph(g):
            body(k)
            if ext:
                ev.record()
        gs.append(g); evs.append(ev)
    return gs, evs


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


gs, _ = make(False)
def plain():
    for g in gs: g.replay()
def waits():
    for g in gs:
        g.replay(); side.wait_stream(main)
        with torch.cuda.stream(side): dummy.add_(1.0)
    main.wait_stream(side)
print(f"4 graphs                               {timeit(plain):.3f} ms")
print(f"4 graphs + wait_stream + side kernel   {timeit(waits):.3f} ms")
try:
    gx, evs = make(True)
    def ext():
        for g, ev in zip(gx, evs):
            g.replay(); side.wait_event(ev)
            with torch.cuda.stream(side): dummy.add_(1.0)
        main.wait_stream(side)
    print(f"4 graphs, external event inside graph  {timeit(ext):.3f} ms")
except Exception as e:
    print("external events:", type(e).__name__, str(e)[:200])
# eager chain for reference: same kernels without graphs, with the waits
def eager_waits():
    for k in range(4):
        body(k); side.wait_stream(main)
        with torch.cuda.stream(side): dummy.add_(1.0)
    main.wait_stream(side)
def eager_plain():
    for k in range(4): body(k)
print(f"eager                                  {timeit(eager_plain):.3f} ms")
print(f"eager + wait_stream + side kernel      {timeit(eager_waits):.3f} ms")
