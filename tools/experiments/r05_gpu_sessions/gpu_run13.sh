#!/bin/bash
# wino4b_kernel (packed transform, V ring 2): producer schedule variants (tools/mb/build_w4.sh W4_DEFS=... W4_TAG=...)
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r05_wino4b_schedule_ab.log
( for rep in 1 2; do for v in cur g1 g1x72 x84 x106; do echo "=== variant $v (rep $rep)"; MB_LIB=tools/mb/libkfnet_w4$v.so MB_BATCH=32 MB_F43_FORM=3 MB_FUSED_ONLY=1 MB_LAYERS=conv1b,conv2b,conv3b,conv4b,conv5 timeout 300 python tools/mb_wino.py 2>&1 | grep -v amdgpu.ids | sed 's/FUSED.*| F(4x4/| F(4x4/'; done; done ) > $L 2>&1
cat $L
