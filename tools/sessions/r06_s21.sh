#!/bin/bash
# round 6, session 21: how the HIP runtime waits (interrupt vs polling) against the frame's two host waits -- alternating processes, 300 frames each
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s21; mkdir -p $O; export PYTHONUNBUFFERED=1
A="--steps 300 --warmup 20 --no-robust --no-cpu-baseline --no-regimes"
: > $O/ab.txt
for i in 1 2; do
  echo "default" >> $O/ab.txt; timeout 200 python bench.py $A 2>> $O/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['step_ms'])" >> $O/ab.txt
  echo "HSA_ENABLE_INTERRUPT=0" >> $O/ab.txt; HSA_ENABLE_INTERRUPT=0 timeout 200 python bench.py $A 2>> $O/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['step_ms'])" >> $O/ab.txt
  echo "ROC_ACTIVE_WAIT_TIMEOUT=1000" >> $O/ab.txt; ROC_ACTIVE_WAIT_TIMEOUT=1000 timeout 200 python bench.py $A 2>> $O/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['step_ms'])" >> $O/ab.txt
done
