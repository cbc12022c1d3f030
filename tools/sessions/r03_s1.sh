#!/bin/bash
# round 3, GPU session 1: F(4x4,3x3) first run on hardware + per-layer A/B + PMC counters of the HEAD GEMM builds
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s1; mkdir -p $O
export PYTHONUNBUFFERED=1
( MSCNN_TEST_WINO_F4=1 timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "f4x4 or wino_f4" -x 2>&1 | tail -30 ) > $O/f4_tests.txt 2>&1
( timeout 400 python tools/bench_layers.py --ab algo=0,3,6 --only conv2_ 2>&1; timeout 400 python tools/bench_layers.py --ab algo=0,3,6 --only conv3_ ; \
  timeout 400 python tools/bench_layers.py --ab algo=0,3,6 --only conv4_ ; timeout 400 python tools/bench_layers.py --ab algo=3,6 --only conv5_ ) > $O/f4_ab.txt 2>&1
# PMC of the GEMM at HEAD (separate passes; --kernel-trace only)
for L in conv3_2 conv4_2 conv5_1; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT \
     --output-format csv -d $O/pmc_$L -- python tools/bench_layers.py --only $L --iters 4 --ab algo=3 > $O/pmc_$L.log 2>&1
  f=$(find $O/pmc_$L -name '*counter_collection.csv' | head -1); python tools/pmc_summary.py $f > $O/pmc_sq_$L.txt 2>&1
  rm -rf $O/pmc_$L
done
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo done > $O/done
