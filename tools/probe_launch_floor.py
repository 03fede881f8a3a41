"""Dependent-kernel latency floor inside a hipGraph: 400 trivial kernels in a chain (each a 1-workgroup cast), replayed."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
Fn = pkg.functional
dev = torch.device("cuda:0")
a = torch.randn(64, device=dev)
b = torch.empty(64, device=dev, dtype=torch.bfloat16)
big = torch.randn(432, 768, device=dev)
bigb = torch.empty(432, 768, device=dev, dtype=torch.bfloat16)
for name, (src, dst) in {"1-workgroup kernel": (a, b), "432x768 cast (162 workgroups)": (big, bigb)}.items():
    Fn.cast_bf16(src, out=dst); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(400):
            Fn.cast_bf16(src, out=dst)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 2000 * 1e3:.2f} us per dependent kernel")
