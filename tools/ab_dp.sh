# the data-parallel launch form on ONE rank with a real RCCL group (bf16 gradient communication): weight-gradient epilogue writing
# the communication buffer (--fuse-comm 1) against fp32 store + cast pass (0); fp32 communication and the single graph beside them
for r in 1 2; do
  for ho in stream host; do
    echo -n "bf16 comm, fused, hand-over $ho: "
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-dist \
      --bf16-comm --handover $ho --no-cpu-baseline --no-roofline --windows 3 2>&1 | grep "timed region" | sed 's/.*done: //'
  done
done
for r in 1; do
  for fc in 0 1; do
    echo -n "bf16 comm, fuse-comm $fc: "
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-dist \
      --bf16-comm --fuse-comm $fc --no-cpu-baseline --no-roofline --windows 3 2>&1 | grep "timed region" | sed 's/.*done: //'
  done
done
echo -n "fp32 comm: "; python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-dist \
      --fp32-comm --no-cpu-baseline --no-roofline --windows 3 2>&1 | grep "timed region" | sed 's/.*done: //'
echo -n "single graph: "; python bench.py --no-cpu-baseline --no-roofline --windows 3 2>&1 | grep "timed region" | sed 's/.*done: //'
