#!/bin/bash
# round-5 GPU call 5: the shipped combination (pointer addressing for the double-buffered scan, compiler-made descriptors for the
# single-buffer forms) in the A/B harness on all four shapes, the whole suite, the driver's command
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -f gpurun_out/conv_error_report.txt
( for a in "256 64" "256 256" "256 32 68 120" "4 64 68 120"; do timeout 300 tools/mb/kalman_mb $a | grep -v "^fuse 5\|^fuse 1\|^fuse 256x8\|^fuse 256x1 nt\|^fuse 256x4 plain"; done ) > gpurun_out/kalman_mb5.log 2>&1
timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/gpu_tests5.log 2>&1
echo "pytest rc $?" >> gpurun_out/gpu_tests5.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver5.json 2> gpurun_out/bench_driver5.err
tail -4 gpurun_out/gpu_tests5.log; grep -c DIFFERS gpurun_out/kalman_mb5.log
