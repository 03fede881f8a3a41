import torch, time
d = torch.device("cuda:0")
a = torch.randn(1, 1 << 22, device=d)
b = torch.randn(1, 1 << 22, device=d)
def work(x):
    for _ in range(50):
        x = torch.cumsum(x, 1) * 1e-3   # few workgroups, long
    return x
s2 = torch.cuda.Stream()
def both_parallel():
    main = torch.cuda.current_stream()
    s2.wait_stream(main)
    with torch.cuda.stream(s2):
        y = work(b)
    x = work(a)
    main.wait_stream(s2)
    return x, y
def both_serial():
    return work(a), work(b)
for name, fn in (("serial", both_serial), ("parallel", both_parallel)):
    for mode in ("eager", "graph"):
        fn(); torch.cuda.synchronize()
        if mode == "graph":
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            run = g.replay
        else:
            run = fn
        run(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5): run()
        torch.cuda.synchronize()
        print(name, mode, (time.perf_counter() - t) / 5 * 1e3, "ms", flush=True)
