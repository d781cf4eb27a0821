cd $GRAFT_REPO_ROOT
echo "== RCCL, one rank under torch.distributed.run"
timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), d['n_gpus'], d['distributed'])"
echo "== gloo, two ranks sharing the GPU (code path only)"
DL_BENCH_BACKEND=gloo timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), d['n_gpus'], d['distributed'], d['config']['sharding'])"
