cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5x
rm -f gpurun_out/conv_error_report.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r5x/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r5x/pytest.log
cp gpurun_out/conv_error_report.txt gpurun_out/r5x/conv_error_report.txt 2>/dev/null
# the test that found the 16-byte-store hazard of conv64_rows_kernel (4 of 10 runs failed before the fix): ten more runs
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 200 python -m pytest tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider -k "batch_independence" 2>&1 | tail -1; done > gpurun_out/r5x/batch_independence_x10.log 2>&1
python __graft_entry__.py smoke 2>&1 | tail -2
bash tools/profile_round.sh r05 > gpurun_out/r5x/profile_round.log 2>&1; echo "profile rc=$?"
tail -c 1500 gpurun_out/prof/r05_bench_driver_command.json
