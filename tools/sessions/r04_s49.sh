#!/bin/bash
# round 4, session 49: full-size parity against the reference's CPU layers for the two other map sizes the ring kernel now takes (8s-768: 768 x 2560; caltech: 480 x 640)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s49; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 175 python -m pytest tests/test_gpu_net.py -m gpu -q -x -k "test_full_size_parity_vs_reference and (caltech or 8s)" 2>&1 | tail -6 ) > $O/tests.txt 2>&1
