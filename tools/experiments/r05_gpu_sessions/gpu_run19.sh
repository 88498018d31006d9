#!/bin/bash
# wino4b_kernel: cache-policy bits on the patch loads (do half-line requests skip the L1 fill of the unused half?)
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r05_wino4b_patch_aux_ab.log
( for rep in 1 2; do for v in aux0 aux1 aux2 aux3; do echo "=== variant $v (rep $rep)"; MB_LIB=tools/mb/libkfnet_w4$v.so MB_BATCH=32 MB_F43_FORM=3 MB_FUSED_ONLY=1 MB_LAYERS=conv2b,conv3b,conv4b timeout 300 python tools/mb_wino.py 2>&1 | grep -v amdgpu.ids | sed 's/FUSED.*| F(4x4/| F(4x4/'; done; done ) > $L 2>&1
cat $L
