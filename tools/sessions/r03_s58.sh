#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s58; mkdir -p $O; export PYTHONUNBUFFERED=1
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driverlike.json 2> $O/bench_driverlike.err
( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench_launcher.json 2> $O/bench_launcher.err
