cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4p
for lib in tools/mb/libkfnet_w4notouch.so tools/mb/libkfnet_w4notouchnb18.so ""; do for ng in 17 18 20 24; do
echo "=== MB_LIB=$lib wino_order=$ng (16 + channel groups adjacent)" >> gpurun_out/r4p/mb_wino.log
MB_LIB=$lib MB_F43_ORDER=$ng MB_BATCH=32 MB_FUSED_ONLY=1 MB_LAYERS=conv2b,conv3b,conv4b,conv5 timeout 300 python tools/mb_wino.py 2>&1 | grep -v amdgpu.ids | sed 's/.*| F(4x4/F(4x4/' >> gpurun_out/r4p/mb_wino.log
done; done
cat gpurun_out/r4p/mb_wino.log
