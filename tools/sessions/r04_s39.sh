#!/bin/bash
# round 4, session 39: chain kernel with 3 / 6 / 8 strips: parity, the 8s-768 net chains on / off
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s39; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "chain" 2>&1 | tail -8 ) > $O/tests.txt 2>&1
{ echo "== 160 tile columns (3 strips), 256 ch"; timeout 120 tools/micro/wino_outin_check 256 192 640 100
  echo "== 320 tile columns (6 strips), 128 ch"; timeout 120 tools/micro/wino_outin_check 128 384 1280 50; } > $O/outin.txt 2>&1
M=kitti_car/mscnn-8s-768-trainval
( timeout 600 python bench.py --model $M --steps 30 --warmup 8 --no-robust --no-alt --no-cpu-baseline ) > $O/bench_8s_chain.json 2> $O/bench_8s_chain.err
( MSCNN_NO_CHAIN=1 timeout 600 python bench.py --model $M --steps 30 --warmup 8 --no-robust --no-alt --no-cpu-baseline ) > $O/bench_8s_nochain.json 2> $O/bench_8s_nochain.err
( timeout 600 python bench.py --model $M --steps 30 --warmup 8 --no-robust --no-alt ) > $O/bench_8s_full.json 2> $O/bench_8s_full.err
