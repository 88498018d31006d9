cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4f/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r4f/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4f/bench.json 2> gpurun_out/r4f/bench.err; echo "bench rc=$?"
tail -c 2300 gpurun_out/r4f/bench.json; tail -5 gpurun_out/r4f/bench.err
