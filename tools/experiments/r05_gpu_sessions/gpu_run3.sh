#!/bin/bash
# round-5 GPU call 3: same-box A/B of the two addressing forms of the scan, split-K of both Winograd kernels in the suite,
# config 2, the real-data block at three chunk sizes
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -f gpurun_out/conv_error_report.txt
( for T in 64 256; do echo "=== pointer addressing (commit 6cf4d79), T=$T"; timeout 300 tools/mb/kalman_mb_ptr 256 $T | grep -v "^fuse"; echo "=== buffer addressing (HEAD), T=$T"; timeout 300 tools/mb/kalman_mb 256 $T | grep -v "^fuse"; done ) > gpurun_out/kalman_mb3.log 2>&1
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/gpu_tests3.log 2>&1
echo "pytest rc $?" >> gpurun_out/gpu_tests3.log
timeout 300 python bench.py --config c2 > gpurun_out/bench_c2_3.json 2> gpurun_out/bench_c2_3.err
for CH in 32 64 128; do
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-alt-modes --no-kalman-roofline --eval-chunk $CH > gpurun_out/bench_eval_chunk$CH.json 2> gpurun_out/bench_eval_chunk$CH.err
done
tail -4 gpurun_out/gpu_tests3.log
