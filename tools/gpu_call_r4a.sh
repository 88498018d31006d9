set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4a/pytest.log 2>&1; echo "pytest rc=$?" 
tail -5 gpurun_out/r4a/pytest.log
MB_K16_ONLY=1 MB_CFGS=9,13,14 timeout 300 python tools/mb_f16.py conv2b conv3a conv3b conv4b conv5 conv6 > gpurun_out/r4a/mb_f16.log 2>&1; echo "mb rc=$?"
cat gpurun_out/r4a/mb_f16.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err; echo "bench rc=$?"
tail -c 2500 gpurun_out/r4a/bench.json; tail -5 gpurun_out/r4a/bench.err
