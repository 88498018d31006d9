export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout 1200 python -m pytest tests -m gpu -q -W ignore > gpurun_out/prof/r02_gpu_tests_final.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" gpurun_out/prof/r02_gpu_tests_final.log | tail -2
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/prof/r02_bench_final.json 2> gpurun_out/prof/bench.err; echo "bench rc=$?"
head -c 260 gpurun_out/prof/r02_bench_final.json; echo
