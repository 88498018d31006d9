cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4r
for lib in "" tools/mb/libkfnet_w4nb18.so tools/mb/libkfnet_w4sp4nb18.so tools/mb/libkfnet_w4sp4nb9.so tools/mb/libkfnet_w4sp2nb18.so; do
echo "=== MB_LIB=$lib" >> gpurun_out/r4r/mb_wino.log
MB_LIB=$lib MB_BATCH=32 MB_FUSED_ONLY=1 MB_LAYERS=conv2b,conv3b,conv4b,conv5 timeout 300 python tools/mb_wino.py 2>&1 | grep -v amdgpu.ids | sed 's/.*| F(4x4/F(4x4/' >> gpurun_out/r4r/mb_wino.log
done
cat gpurun_out/r4r/mb_wino.log
( for t in a c; do echo "=== prof $t (a: super-step 20, c: 21): touches in 12 pieces of 16 lanes every 11 slots, B ring 18"; timeout 120 tools/mb/wino4_prof_$t 16 60 80 1024 1024 | tail -3; done ) > gpurun_out/r4r/wino4_prof.log 2>&1
cat gpurun_out/r4r/wino4_prof.log
