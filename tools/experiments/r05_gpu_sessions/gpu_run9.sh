#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( echo "=== shipped library"; timeout 600 python tools/debug_conv64.py 40 ) > gpurun_out/debug_conv64b.log 2>&1
cat gpurun_out/debug_conv64b.log | grep -v amdgpu.ids | head -80
