#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=gpurun_out/r05_conv64_store_hazard.log
( echo "=== shipped library (KFN_STORE_PAD=7: s_nop 7 behind every 16-byte buffer store)"; timeout 900 python tools/debug_conv64.py 100
  for p in -1 0 1; do echo "=== KFN_STORE_PAD=$p"; MB_LIB=tools/mb/libkfnet_pad$p.so timeout 600 python tools/debug_conv64.py 100 quick; done ) > $L 2>&1
grep -v amdgpu.ids $L | grep "===\|launches differ"
( for i in 1 2 3 4 5 6 7 8 9 10; do timeout 300 python -m pytest tests/test_gpu_e2e.py -q -x -k test_config5_batch_independence_at_full_size 2>&1 | tail -1; done ) > gpurun_out/batch_independence_x10b.log 2>&1
cat gpurun_out/batch_independence_x10b.log
