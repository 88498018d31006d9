cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4v
for o in 1 2 17 18 20; do
echo "=== wino_order=$o" >> gpurun_out/r4v/mb_s2.log
MB_WINO_ORDER=$o MB_BATCH=32 timeout 300 python tools/mb_s2.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4v/mb_s2.log
done
cat gpurun_out/r4v/mb_s2.log
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "winograd_s2" 2>&1 | tail -2
