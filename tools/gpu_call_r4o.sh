cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4o
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "f43" 2>&1 | tail -2
for lib in "" tools/mb/libkfnet_w4notouch.so tools/mb/libkfnet_w4t74nb12.so tools/mb/libkfnet_w4t74nb18.so tools/mb/libkfnet_w4t4.so tools/mb/libkfnet_w4t110.so; do
echo "=== MB_LIB=$lib" >> gpurun_out/r4o/mb_wino.log
MB_LIB=$lib MB_BATCH=32 MB_FUSED_ONLY=1 MB_LAYERS=conv1b,conv2b,conv3b,conv4b,conv5 timeout 300 python tools/mb_wino.py 2>&1 | grep -v amdgpu.ids | sed 's/| two-kernel.*(nan TF) |/|/' >> gpurun_out/r4o/mb_wino.log
done
cat gpurun_out/r4o/mb_wino.log
( for t in a c; do echo "=== prof variant $t (a: super-step 20, c: super-step 21), line touches on"; timeout 120 tools/mb/wino4_prof_$t 16 60 80 1024 1024 | tail -3; done ) > gpurun_out/r4o/wino4_prof.log 2>&1
cat gpurun_out/r4o/wino4_prof.log
