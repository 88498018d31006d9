cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
for lib in "" tools/mb/libkfnet_w4x88.so tools/mb/libkfnet_w4dbg16.so tools/mb/libkfnet_w4dbg32.so tools/mb/libkfnet_w4dbg48.so tools/mb/libkfnet_w4dbg2.so; do
  echo "=== MB_LIB=$lib" >> gpurun_out/r4g/mb_wino.log
  MB_LIB=$lib MB_BATCH=32 MB_FUSED_ONLY=1 MB_LAYERS=conv2b,conv3b,conv4b,conv6 timeout 300 python tools/mb_wino.py 2>&1 | grep -v amdgpu.ids | sed 's/| two-kernel.*(nan TF) |/|/' >> gpurun_out/r4g/mb_wino.log
done
echo "=== batch 20" >> gpurun_out/r4g/mb_wino.log
MB_BATCH=20 MB_FUSED_ONLY=1 MB_LAYERS=conv2b,conv3b,conv4b,conv5,conv6 timeout 300 python tools/mb_wino.py 2>&1 | grep -v amdgpu.ids | sed 's/| two-kernel.*(nan TF) |/|/' >> gpurun_out/r4g/mb_wino.log
cat gpurun_out/r4g/mb_wino.log
