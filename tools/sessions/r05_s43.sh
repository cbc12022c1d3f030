#!/bin/bash
# round 5, session 43: cls_pred / bbox_pred kernel at 2 rows per workgroup: back to back and inside the net
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s43; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "inner_product_small_n or test_inner_product" 2>&1 | tail -3 ) > $O/tests.txt 2>&1
timeout 200 python tools/debug/ip_rows_ab.py > $O/ip_rows.txt 2>&1
for rows in 2 4 2 4; do
  timeout 200 python -c "
import sys, runpy
from mscnn_amd import hipapi
hipapi.debug_inner_product_rows($rows)
sys.argv = ['bench.py', '--steps', '100', '--warmup', '10', '--no-robust', '--no-cpu-baseline', '--layers']
runpy.run_path('bench.py', run_name='__main__')
" > $O/bench_rows$rows.json 2> $O/layers_rows$rows.txt
  grep '^{' $O/bench_rows$rows.json | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('rows', $rows, r['value'], r['ms_per_step'])" >> $O/ip_rows.txt
  grep -E "^cls_pred|^bbox_pred" $O/layers_rows$rows.txt | cut -c1-100 >> $O/ip_rows.txt
done
