#!/bin/bash
# round-5 GPU call 2: Kalman kernels with buffer addressing, F(4x4) split-K, the suite, config 2, the driver's command
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -f gpurun_out/conv_error_report.txt
( timeout 300 tools/mb/kalman_mb 256 64 ; timeout 300 tools/mb/kalman_mb 256 256 ; timeout 200 tools/mb/kalman_mb 256 32 68 120 ; timeout 200 tools/mb/kalman_mb 4 64 68 120 ) > gpurun_out/kalman_mb2.log 2>&1
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/gpu_tests2.log 2>&1
echo "pytest rc $?" >> gpurun_out/gpu_tests2.log
timeout 300 python bench.py --config c2 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
timeout 300 python bench.py --config c2 --graph-option x=1 > /dev/null 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver2.json 2> gpurun_out/bench_driver2.err
( cd /tmp && timeout 120 rocprofv3 -L 2>/dev/null | grep -i -E "^\s*(gpu|Name|.*TA_|.*TCP_|.*SQ_INSTS_VMEM|.*SQ_WAIT_INST)" | head -120 ) > gpurun_out/counters_list.txt 2>&1
tail -5 gpurun_out/gpu_tests2.log
tail -c 600 gpurun_out/bench_c2.json
