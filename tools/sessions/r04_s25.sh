#!/bin/bash
# round 4, session 25: the strict-metric table (direct kernel / Winograd / reference CPU layer vs float64 on vgg_like activations)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s25; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python tools/strict_metric_table.py ) > $O/strict_metric.txt 2> $O/strict_metric.err
( timeout 600 python tools/strict_metric_table.py --style he ) > $O/strict_metric_he.txt 2>> $O/strict_metric.err
