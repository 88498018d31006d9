set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r02b
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -W ignore -k "winograd_fused" > gpurun_out/r02b/tests_fused.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r02b/tests_fused.log
timeout 600 python tools/mb_wino.py > gpurun_out/r02b/mb_wino.log 2>&1; echo "mb rc=$?"
cat gpurun_out/r02b/mb_wino.log
