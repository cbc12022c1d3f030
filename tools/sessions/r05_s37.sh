#!/bin/bash
# round 5, session 37: the driver's own N = 1 command (20 steps + 5 warm-up, every leg on), twice; and under the launcher at world 1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s37; mkdir -p $O; export PYTHONUNBUFFERED=1
for i in 1 2; do ( time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/driver_$i.json 2> $O/driver_$i.err; done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-robust > $O/launcher.json 2> $O/launcher.err
