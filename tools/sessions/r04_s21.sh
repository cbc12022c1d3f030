#!/bin/bash
# round 4, session 21: fused F(4x4,3x3) output -> input transform, all strips of a tile row in one workgroup (S x RW waves)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s21; mkdir -p $O
B=tools/micro/wino_outin_check
{
echo "== conv4_1 -> conv4_2 (512 x 72 x 240)"; timeout 120 $B 512 72 240 100
echo "== conv3_1 -> conv3_2 (256 x 144 x 480), default chunks"; timeout 120 $B 256 144 480 100
echo "== conv3, whole height"; timeout 120 $B 256 144 480 100 0 36
echo "== conv3, chunks of 9"; timeout 120 $B 256 144 480 100 0 9
echo "== conv2_1 -> conv2_2 (128 x 288 x 960), default chunks"; timeout 120 $B 128 288 960 50
echo "== conv2, whole height"; timeout 120 $B 128 288 960 50 0 72
echo "== conv2, chunks of 18"; timeout 120 $B 128 288 960 50 0 18
echo "== conv2, chunks of 9"; timeout 120 $B 128 288 960 50 0 9
echo "== odd: 24 x 40 x 72"; timeout 120 $B 24 40 72 20
echo "== odd: 7 x 28 x 480 (two strips)"; timeout 120 $B 7 28 480 20
echo "== odd: 5 x 20 x 800 (four strips of 50)"; timeout 120 $B 5 20 800 20
} > $O/outin.txt 2>&1
