#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r05_wino4b_packed_ab.log
( for rep in 1 2; do for v in base p3 p2 b2; do echo "=== variant $v (rep $rep)"; MB_LIB=tools/mb/libkfnet_w4$v.so MB_BATCH=32 MB_F43_FORM=3 MB_FUSED_ONLY=1 MB_LAYERS=conv1b,conv2b,conv3b,conv4b,conv5,conv6 timeout 300 python tools/mb_wino.py 2>&1 | grep -v amdgpu.ids; done; done ) > $L 2>&1
cat $L
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "f43 or winograd" 2>&1 | tail -3
