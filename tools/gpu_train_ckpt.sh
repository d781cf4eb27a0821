#!/bin/bash
# Train the walking policy the contact-rich benchmark line runs: examples/train_ppo.py with the reference's budget
# (8 M env-steps), a few seeds; checkpoints + logs under gpurun_out/ckpt/ (the one that reaches the 3000-step limit is then condensed
# into drloco_amd/data/ by tools/pack_walking_ckpt.py).
out=gpurun_out/ckpt; mkdir -p $out
for s in ${1:-1 2 3}; do
  timeout 300 python examples/train_ppo.py --mio 8 --seed $s --save $out/s$s > $out/train_s$s.log 2>&1
  tail -4 $out/train_s$s.log
done
