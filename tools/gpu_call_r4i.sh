cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r4i/pytest.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r4i/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4i/bench.json 2> gpurun_out/r4i/bench.err; echo "bench rc=$?"
tail -c 2000 gpurun_out/r4i/bench.json; tail -3 gpurun_out/r4i/bench.err
