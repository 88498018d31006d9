cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4u
MB_CFGS=14 timeout 300 python tools/mb_f16.py conv3b conv4b conv5 2>&1 | grep -v amdgpu.ids > gpurun_out/r4u/mb_f16.log
cat gpurun_out/r4u/mb_f16.log
