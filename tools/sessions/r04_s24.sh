#!/bin/bash
# round 4, session 24: pool-only forwards on the direct kernel (conv1_2) too: tests, bench A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s24; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "chain or pool" 2>&1 | tail -15 ) > $O/tests_ops.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_net.py -q -x -k "chains or fusion or partial or precision" 2>&1 | tail -15 ) > $O/tests_net.txt 2>&1
for i in 1 2; do
  ( timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-alt --no-robust ) > $O/bench_chain_$i.json 2> $O/bench_chain_$i.err
  ( MSCNN_NO_CHAIN=1 timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-alt --no-robust ) > $O/bench_nochain_$i.json 2> $O/bench_nochain_$i.err
done
( timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-alt --no-robust --layers ) > $O/layers_chain.json 2> $O/layers_chain.txt
