#!/bin/bash
# round-5 GPU call 4: scan with uniform descriptors (no waterfall loops) against the pointer form on one box, the whole suite
# (block-cyclic sharding, split-K of both kernels), the real-data block at small chunk sizes, the driver's command
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -f gpurun_out/conv_error_report.txt
( for T in 64 256; do echo "=== pointer addressing (commit 6cf4d79), T=$T"; timeout 300 tools/mb/kalman_mb_ptr 256 $T | grep -v "^fuse" | grep -E "^#|^r4|1024x5 D=5 nt|768x7  D=7 nt"; echo "=== buffer addressing, uniform descriptors (HEAD), T=$T"; timeout 300 tools/mb/kalman_mb 256 $T | grep -v "^fuse"; done; timeout 200 tools/mb/kalman_mb 256 32 68 120 | grep -v "^fuse"; timeout 200 tools/mb/kalman_mb 4 64 68 120 | grep -v "^fuse" ) > gpurun_out/kalman_mb4.log 2>&1
timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/gpu_tests4.log 2>&1
echo "pytest rc $?" >> gpurun_out/gpu_tests4.log
for CH in 16 24 32; do
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-alt-modes --no-kalman-roofline --eval-chunk $CH > gpurun_out/bench_eval_chunk$CH.json 2> gpurun_out/bench_eval_chunk$CH.err
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver4.json 2> gpurun_out/bench_driver4.err
tail -4 gpurun_out/gpu_tests4.log
