#!/bin/bash
# full GPU suite + the distributed code paths on the 1-GPU box (RCCL with one rank; two gloo ranks sharing the GPU)
TAG=${1:-r02h}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "flips|reward error|passed|failed|rc=" $OUT/pytest.log | tail -12
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 2 --no-cpu-baseline > $OUT/bench_rccl1.json 2> $OUT/bench_rccl1.err
python3 -c "import json; d=json.load(open('$OUT/bench_rccl1.json')); print('rccl 1 rank', round(d['value']/1e6,2), d['distributed'])" || tail -5 $OUT/bench_rccl1.err
DL_BENCH_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 1 --envs-per-gpu 2048 --no-cpu-baseline > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err
python3 -c "import json; d=json.load(open('$OUT/bench_gloo2.json')); print('gloo 2 ranks sharing the GPU', round(d['value']/1e6,2), d['distributed'])" || tail -5 $OUT/bench_gloo2.err
timeout 900 python3 examples/train_ppo.py --mio 0.4 > $OUT/train1.log 2>&1; tail -3 $OUT/train1.log
DL_BENCH_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 examples/train_ppo.py --mio 0.4 > $OUT/train2.log 2>&1; tail -3 $OUT/train2.log
