#!/bin/bash
# round 6, session 8: A/B of the final stage's modes in one process each, alternating: sync | pipelined (copy in the compute stream) | pipelined (copy on its own stream)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s8; mkdir -p $O; export PYTHONUNBUFFERED=1
A="--steps 300 --warmup 20 --no-robust --no-cpu-baseline --no-regimes"
: > $O/ab.jsonl
for i in 1 2; do
  timeout 200 python bench.py $A --detect-mode sync >> $O/ab.jsonl 2>> $O/ab.err
  timeout 200 python bench.py $A --detect-mode pipelined >> $O/ab.jsonl 2>> $O/ab.err
  MSCNN_DETECT_COPY_STREAM=1 timeout 200 python bench.py $A --detect-mode pipelined >> $O/ab.jsonl 2>> $O/ab.err
done
timeout 120 python bench.py --steps 300 --warmup 20 --no-robust --no-cpu-baseline --no-regimes --detect-mode pipelined --dump-steps > /dev/null 2> $O/steps_pipelined.txt
