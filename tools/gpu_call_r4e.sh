cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "f43" > gpurun_out/r4e/pytest_f43.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r4e/pytest_f43.log
for lib in "" tools/mb/libkfnet_w4nb18.so tools/mb/libkfnet_w4g1.so tools/mb/libkfnet_w4g1nb18.so tools/mb/libkfnet_w4g1nb18x80.so tools/mb/libkfnet_w4dbg1.so tools/mb/libkfnet_w4dbg2.so; do
  echo "=== MB_LIB=$lib" >> gpurun_out/r4e/mb_wino.log
  MB_LIB=$lib MB_BATCH=32 MB_FUSED_ONLY=1 MB_LAYERS=conv3b,conv4b,conv5 timeout 300 python tools/mb_wino.py 2>&1 | grep -v amdgpu.ids | sed 's/| two-kernel.*(nan TF) |/|/' >> gpurun_out/r4e/mb_wino.log
done
cat gpurun_out/r4e/mb_wino.log
