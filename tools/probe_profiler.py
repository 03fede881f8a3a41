"""does torch.profiler see HIP kernels (also those replayed from a hipGraph) on this stack?"""
import torch
from torch.profiler import profile, ProfilerActivity
x = torch.randn(1 << 20, device="cuda")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    y = x * 2
torch.cuda.synchronize()
with torch.cuda.graph(g):
    y = x * 2 + 1
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3):
        g.replay()
    z = x.sin()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=8))
