#!/bin/bash
# round 6, session 24: the N > 1 code path at the one world size this box allows, under the launcher (real RCCL, pipelined gather, per-rank stats, placement)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s24; mkdir -p $O; export PYTHONUNBUFFERED=1
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 200 --warmup 20 > $O/launched.json 2> $O/launched.err
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 200 --warmup 20 --tiles whole > $O/launched_whole.json 2>> $O/launched.err
