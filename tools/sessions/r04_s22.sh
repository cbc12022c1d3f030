#!/bin/bash
# round 4, session 22: convolution chains (fused output -> input transform) through the C ABI and the Net; bench A/B chains on / off
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s22; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "chain or roipool_pair or winograd_f4" 2>&1 | tail -15 ) > $O/tests_ops.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_net.py -q -x -k "chains or deferred or default_flow or partial" 2>&1 | tail -15 ) > $O/tests_net.txt 2>&1
for i in 1 2; do
  ( timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-alt ) > $O/bench_chain_$i.json 2> $O/bench_chain_$i.err
  ( MSCNN_NO_CHAIN=1 timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-alt ) > $O/bench_nochain_$i.json 2> $O/bench_nochain_$i.err
done
