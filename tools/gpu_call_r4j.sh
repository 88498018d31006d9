cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
MB_K16_ONLY=1 timeout 300 python tools/mb_f16.py conv1b 2>&1 | grep -v amdgpu.ids > gpurun_out/r4j/mb_f16.log
cat gpurun_out/r4j/mb_f16.log
