#!/bin/bash
# round-5 GPU call 6: where does the batch-independence difference of config 5 come from, and is it stable?
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python tools/debug_c5_batch.py > gpurun_out/debug_c5_batch.log 2>&1
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider -k "batch_independence or config5_shape" 2>&1 | tail -2; done >> gpurun_out/debug_c5_batch.log 2>&1
cat gpurun_out/debug_c5_batch.log | tail -40
