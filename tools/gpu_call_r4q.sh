cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4q
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r4q/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r4q/pytest.log
bash tools/profile_round.sh r04 > gpurun_out/r4q/profile_round.log 2>&1; echo "profile rc=$?"
tail -30 gpurun_out/r4q/profile_round.log
