#!/bin/bash
# round 5, session 28: A/B nontemporal loads of M / stores of V in the F(4x4,3x3) transform kernels (dev library interposed with LD_PRELOAD), alternating
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s28; mkdir -p $O; export PYTHONUNBUFFERED=1
for i in 1 2; do
  timeout 200 python bench.py --steps 60 --warmup 10 --no-robust --no-cpu-baseline --layers > $O/base_$i.json 2> $O/base_layers_$i.txt
  LD_PRELOAD=$GRAFT_REPO_ROOT/tools/micro/libmscnn_hip_nt.so timeout 200 python bench.py --steps 60 --warmup 10 --no-robust --no-cpu-baseline --layers > $O/nt_$i.json 2> $O/nt_layers_$i.txt
done
