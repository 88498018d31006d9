export TMPDIR=/tmp
cp kfnet_amd/libkfnet_hip.so /tmp/kfn_keep.so
for i in 1 2; do for L in gpurun_lib_old.so gpurun_lib_new.so; do
  cp $L kfnet_amd/libkfnet_hip.so
  echo "== $L"
  MB_LAYERS=conv2b,conv3b,conv4b,conv6 MB_FUSED_ONLY=1 MB_BATCH=16 python tools/mb_wino.py 2>&1 | grep -v amdgpu | cut -c1-62
done; done
cp /tmp/kfn_keep.so kfnet_amd/libkfnet_hip.so
