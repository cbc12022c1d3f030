#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s61; mkdir -p $O; export PYTHONUNBUFFERED=1
timeout 900 python bench.py --layers > $O/bench.json 2> $O/bench_layers.txt
( timeout 300 python -m pytest tests/test_gpu_net.py -q -k "full_size_parity_vs_reference or layerwise" 2>&1 | tail -3 ) > $O/net.txt 2>&1
