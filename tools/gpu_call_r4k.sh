cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
for o in 1 2; do
echo "=== wino_order=$o (1 = tile blocks fastest, 2 = channel groups fastest)" >> gpurun_out/r4k/mb_wino.log
MB_WINO_ORDER=$o MB_BATCH=32 MB_FUSED_ONLY=1 MB_LAYERS=conv1b,conv2b,conv3b,conv4b,conv5,conv6 timeout 300 python tools/mb_wino.py 2>&1 | grep -v amdgpu.ids | sed 's/| two-kernel.*(nan TF) |/|/' >> gpurun_out/r4k/mb_wino.log
done
cat gpurun_out/r4k/mb_wino.log
