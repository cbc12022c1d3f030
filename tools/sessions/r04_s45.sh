#!/bin/bash
# round 4, session 45: last sanity at HEAD (host library gained a getter since the full run): smoke, the chain / default-flow net tests, a short bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s45; mkdir -p $O; export PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/smoke.txt
( timeout 600 python -m pytest tests/test_gpu_net.py tests/test_cabi.py -m gpu -q -k "chains or default_flow or deferred or boundary or reference_style" 2>&1 | tail -5 ) > $O/tests.txt 2>&1
( timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-alt --no-robust ) > $O/bench.json 2> $O/bench.err
