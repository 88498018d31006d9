#!/bin/bash
# wino4b_kernel: two channels + half the patch rows per producer lane (KFN_W4B_PAIR), correctness then A/B
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_ops.py -q -x -k "f43 or border" -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror" | tail -5
L=gpurun_out/r05_wino4b_pair_ab.log
( for rep in 1 2; do for v in pair0 pair1; do echo "=== variant $v (rep $rep)"; MB_LIB=tools/mb/libkfnet_w4$v.so MB_BATCH=32 MB_F43_FORM=3 MB_FUSED_ONLY=1 MB_LAYERS=conv1b,conv2b,conv3b,conv4b,conv5,conv6 timeout 300 python tools/mb_wino.py 2>&1 | grep -v amdgpu.ids | sed 's/FUSED.*| F(4x4/| F(4x4/'; done; done ) > $L 2>&1
cat $L
