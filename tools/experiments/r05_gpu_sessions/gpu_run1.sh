#!/bin/bash
# round-5 GPU call 1: Kalman A/B micro-benchmark, the full GPU suite (not -x: every failure of the new error model shows),
# the driver's bench command
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -f gpurun_out/conv_error_report.txt
( timeout 300 tools/mb/kalman_mb 256 64 ; timeout 300 tools/mb/kalman_mb 256 256 ; timeout 200 tools/mb/kalman_mb 256 32 68 120 ; timeout 200 tools/mb/kalman_mb 4 64 68 120 ) > gpurun_out/kalman_mb.log 2>&1
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/gpu_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/gpu_tests.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err
tail -5 gpurun_out/gpu_tests.log
tail -c 1500 gpurun_out/bench_driver.json
