# 2 000 steps of the data-parallel launch form on one rank (real RCCL group, bf16 gradient communication): the default form
# (bf16 gradients from the weight-gradient epilogue, hand-over through the host) against fp32 store + cast pass with stream waits --
# the two must end at the same loss, bit for bit
for cfg in "--fuse-comm 1 --handover host" "--fuse-comm 0 --handover stream"; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --force-dist \
    --bf16-comm $cfg --steps 400 --windows 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', d['ms_per_step'], d['config']['first_step_loss'], d['config']['final_loss'])"
done
