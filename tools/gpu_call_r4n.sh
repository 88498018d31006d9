cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4n
for lib in "" tools/mb/libkfnet_w4nb18.so; do for o in 1 2; do
echo "=== MB_LIB=$lib wino_order=$o (1 = tile blocks fastest, 2 = channel groups fastest)" >> gpurun_out/r4n/mb_wino.log
MB_LIB=$lib MB_WINO_ORDER=$o MB_BATCH=32 MB_FUSED_ONLY=1 MB_LAYERS=conv2b,conv3b,conv4b,conv5 timeout 300 python tools/mb_wino.py 2>&1 | grep -v amdgpu.ids | sed 's/| two-kernel.*(nan TF) |/|/' >> gpurun_out/r4n/mb_wino.log
done; done
cat gpurun_out/r4n/mb_wino.log
