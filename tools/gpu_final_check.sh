cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4w
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r4w/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r4w/pytest.log
python __graft_entry__.py smoke 2>&1 | tail -2
bash tools/profile_round.sh r04 > gpurun_out/r4w/profile_round.log 2>&1; echo "profile rc=$?"
tail -c 1500 gpurun_out/prof/r04_bench_driver_command.json
