#!/bin/bash
# round 3, last session: the whole GPU suite + smoke at HEAD, then the final measurements (as r03_s49.sh)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s54; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/gpu_all.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 900 python bench.py --layers > $O/bench.json 2> $O/bench_layers.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-robust > $O/bench_prof.json 2> $O/bench_prof.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/stats
for M in kitti_car/mscnn-8s-768-trainval kitti_ped_cyc/mscnn-7s-576-2x caltech/mscnn-7s-480; do
  timeout 600 python bench.py --model $M --steps 30 --warmup 8 --no-robust > $O/bench_$(basename $M).json 2> $O/bench_$(basename $M).err
done
timeout 300 python bench.py --model caltech/mscnn-7s-480 --dtype f16 --steps 30 --warmup 8 --no-robust > $O/bench_caltech_f16.json 2> $O/bench_caltech_f16.err
