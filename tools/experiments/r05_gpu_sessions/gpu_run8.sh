#!/bin/bash
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python tools/debug_c5_batch.py 30 > gpurun_out/debug_c5_batch2.log 2>&1
tail -30 gpurun_out/debug_c5_batch2.log
