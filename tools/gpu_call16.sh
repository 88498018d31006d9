export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py -m gpu -q -k "window_fc" -x 2>&1 | tail -2
python -m pytest tests/test_gpu_e2e.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -2
bash tools/gpu_call12.sh
