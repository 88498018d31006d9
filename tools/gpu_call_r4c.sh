set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "f43" > gpurun_out/r4c/pytest_f43.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r4c/pytest_f43.log
MB_BATCH=32 MB_FUSED_ONLY=1 MB_LAYERS=conv2b,conv3b,conv4b,conv5,conv6 timeout 300 python tools/mb_wino.py > gpurun_out/r4c/mb_wino.log 2>&1; echo "mb rc=$?"
cat gpurun_out/r4c/mb_wino.log
