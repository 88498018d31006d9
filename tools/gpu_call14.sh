export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py -m gpu -q -k "winograd_fused" -x -s 2>&1 | grep -E "f16:|passed|failed|Error|error" | tail -9
MB_F16=1 MB_FUSED_ONLY=1 MB_BATCH=16 python tools/mb_wino.py 2>&1 | grep -v amdgpu | cut -c1-75
python -m pytest tests/test_gpu_e2e.py -m gpu -q -k "config5" -x -s 2>&1 | grep -E "fp16-operand|passed|failed|Error" | tail -5
timeout 400 python bench.py --config c5 --no-cpu-baseline > gpurun_out/c5.json 2> gpurun_out/c5.err; echo "c5 rc=$?"; head -c 700 gpurun_out/c5.json; echo; tail -3 gpurun_out/c5.err
