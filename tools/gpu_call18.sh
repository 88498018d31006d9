set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
python __graft_entry__.py smoke 2>&1 | tail -2
timeout 1500 bash tools/profile_round.sh r02 > gpurun_out/prof/profile_round.log 2>&1
timeout 400 python bench.py --config c5 > gpurun_out/prof/r02_bench_c5.json 2> gpurun_out/prof/c5.err
head -c 300 gpurun_out/prof/r02_bench_final.json; echo
head -c 300 gpurun_out/prof/r02_bench_c5.json; echo
