cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
MB_BATCH=32 MB_FUSED_ONLY=1 MB_LAYERS=conv1b,feat5,feat3 timeout 300 python tools/mb_wino.py 2>&1 | grep -v amdgpu.ids | sed 's/| two-kernel.*(nan TF) |/|/' > gpurun_out/r4h/mb_wino.log
cat gpurun_out/r4h/mb_wino.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kalman-roofline --no-extra-configs > gpurun_out/r4h/bench.json 2> gpurun_out/r4h/bench.err; echo "bench rc=$?"
tail -c 1200 gpurun_out/r4h/bench.json; tail -3 gpurun_out/r4h/bench.err
