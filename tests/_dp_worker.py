"""Worker of tests/test_model_gpu.py::test_two_rank_data_parallel_step: one rank of a 2-rank data-parallel job on ONE GPU
(gloo moves the CUDA gradient pieces; RCCL refuses two ranks on one device).  Usage: python _dp_worker.py RANK PORT OUTDIR MODE [c1|c2]
(c2 = BASELINE configs[2]'s per-rank workload: 96^3, hidden 768, 4 classes, bf16, batch 2 per rank)"""
import importlib
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rank, port, outdir, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
size = sys.argv[5] if len(sys.argv) > 5 else "c1"
pkg = importlib.import_module("3dmedicalimagesegmentation_amd")
from tools.synthetic import synthetic_volume

C1 = dict(in_channels=1, out_channels=2, img_size=(32, 32, 32), feature_size=16, hidden_size=128, mlp_dim=512, num_heads=4,
          pos_embed="perceptron", norm_name="instance", res_block=True)
C2 = dict(in_channels=1, out_channels=4, img_size=(96, 96, 96), feature_size=16, hidden_size=768, mlp_dim=3072, num_heads=12,
          pos_embed="perceptron", norm_name="instance", res_block=True)
cfg, S, ncls, lr = (C2, 96, 4, 1e-4) if size == "c2" else (C1, 32, 2, 1e-3)
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
dev = torch.device("cuda:0")
if size == "c5":
    # BASELINE configs[4] in its data-parallel form at the plumbing geometry: per rank a [4, 1, 32^3] batch (2 volumes x 2 views),
    # one "feat" pass (loss on enc4, staged backward, per-pass all-reduce) and one "recon" pass (encoder frozen, loss on the logits,
    # one backward pass whose gradient runs are the communication pieces) per round, as bench.py --config c5 --gpus N builds them
    torch.manual_seed(11)
    m = pkg.UNETR(**C1).to(dev)
    m.precision = "bf16"
    flat = m.use_flat_buffers()
    opt = pkg.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5, flat=flat)
    x, _ = synthetic_volume(8, 1, 32, 2, seed=77)        # joint batch [r0.f1, r1.f1, r0.f2, r1.f2]: this rank's views 2r, 2r+1 of each half
    xs = torch.cat([x[2 * rank:2 * rank + 2], x[4 + 2 * rank:4 + 2 * rank + 2]]).to(dev)
    feat = pkg.TrainStep(m, None, opt, xs, None, use_graph=mode == "graph", data_parallel=True, warmup=1, comm_dtype=torch.float32,
                         loss_fn=lambda e, l: pkg.ranking_loss(e, 2, 0, 0.1, kind="ranking"))
    recon = pkg.TrainStep(m, None, opt, xs, None, use_graph=mode == "graph", data_parallel=True, warmup=1, comm_dtype=torch.float32,
                          loss_fn=lambda e, l: pkg.ranking_loss(l, 4, 3, 0.1, kind="ranking"), freeze_encoder=True)
    for _ in range(2):
        feat.run()
        recon.run()
    torch.cuda.synchronize()
    torch.save({"param": flat["param"].cpu(), "loss": (float(feat.loss), float(recon.loss)), "steps": (feat.eager_steps, recon.eager_steps),
                "pieces": (feat.pieces, recon.pieces)}, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0)
torch.manual_seed(11)                                   # same initial weights on both ranks
m = pkg.UNETRLogits(**cfg).to(dev)
m.precision = "bf16"
flat = m.use_flat_buffers()
opt = pkg.AdamW(m.parameters(), lr=lr, weight_decay=1e-5, flat=flat)
crit = pkg.DiceCELoss(to_onehot_y=True, softmax=True)
x, y = synthetic_volume(4, 1, S, ncls, seed=77)          # the global batch; this rank's shard = samples 2r, 2r+1
xs, ys = x[2 * rank:2 * rank + 2].to(dev), y[2 * rank:2 * rank + 2].to(dev)
step = pkg.TrainStep(m, crit, opt, xs, ys, use_graph=mode == "graph", data_parallel=True, warmup=1,
                     comm_dtype=torch.float32)
for _ in range(2):
    step.run()
torch.cuda.synchronize()
torch.save({"param": flat["param"].cpu(), "loss": float(step.loss), "steps": step.eager_steps}, os.path.join(outdir, f"rank{rank}.pt"))
dist.barrier()
dist.destroy_process_group()
