set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
for lib in "" tools/mb/libkfnet_pin0.so tools/mb/libkfnet_bar4.so; do
  echo "=== MB_LIB=$lib" >> gpurun_out/r4b/mb_f16.log
  MB_LIB=$lib MB_K16_ONLY=1 MB_CFGS=9,14 timeout 300 python tools/mb_f16.py conv2b conv3b conv4b conv5 >> gpurun_out/r4b/mb_f16.log 2>&1
done
cat gpurun_out/r4b/mb_f16.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kalman-roofline > gpurun_out/r4b/bench.json 2> gpurun_out/r4b/bench.err; echo "bench rc=$?"
tail -c 1800 gpurun_out/r4b/bench.json; tail -5 gpurun_out/r4b/bench.err
