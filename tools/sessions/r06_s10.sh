#!/bin/bash
# round 6, session 10: what whole-tile scheduling of the plane GEMM costs (the N > 1 default), alternating processes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s10; mkdir -p $O; export PYTHONUNBUFFERED=1
A="--steps 200 --warmup 20 --no-robust --no-cpu-baseline --no-regimes"
: > $O/ab.jsonl
for i in 1 2; do
  timeout 200 python bench.py $A --tiles split >> $O/ab.jsonl 2>> $O/ab.err
  timeout 200 python bench.py $A --tiles whole >> $O/ab.jsonl 2>> $O/ab.err
done
