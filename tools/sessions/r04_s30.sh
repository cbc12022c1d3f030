#!/bin/bash
# round 4, session 30: conv1_2 epilogue ablations (dev builds of conv.hip: no bias loads / no pooling pass)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s30; mkdir -p $O; export PYTHONUNBUFFERED=1
{
for rep in 1 2; do
echo "== product build, pooled only"; timeout 300 python tools/bench_layers.py --only conv1_2 --iters 60 --pool only
echo "== no bias loads, pooled only"; timeout 300 python tools/bench_layers.py --only conv1_2 --iters 60 --pool only --lib tools/micro/libmscnn_hip_NOBIAS.so
echo "== product build, y + pooled"; timeout 300 python tools/bench_layers.py --only conv1_2 --iters 60 --pool both
echo "== no pooling pass, y only"; timeout 300 python tools/bench_layers.py --only conv1_2 --iters 60 --pool both --lib tools/micro/libmscnn_hip_NOPOOL.so
done
} > $O/ablate.txt 2>&1
