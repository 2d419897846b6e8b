cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -4
BENCH_BACKEND=gloo BENCH_SHARE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 3 --repeats 2 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-600
