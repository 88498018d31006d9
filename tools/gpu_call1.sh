set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r02a
timeout 1200 python -m pytest tests -m gpu -q -W ignore --durations=15 > gpurun_out/r02a/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/tests.log
tail -40 gpurun_out/r02a/tests.log
timeout 600 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --config c2 > gpurun_out/r02a/c2.json 2> gpurun_out/r02a/c2.err; echo "c2 rc=$?"
timeout 400 python bench.py --config c5 > gpurun_out/r02a/c5.json 2> gpurun_out/r02a/c5.err; echo "c5 rc=$?"
R=$(pwd)
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/r02a/k$C -- python $R/tools/kalman_roofline.py > $R/gpurun_out/r02a/kalman_$C.json 2> $R/gpurun_out/r02a/k$C.err )
done
finddb() { find "$1" -name '*.db' | head -1; }
cp profiles/pmc_traffic.json gpurun_out/r02a/pmc_traffic.json
python tools/pmc_traffic.py "$(finddb gpurun_out/r02a/kFETCH_SIZE)" "$(finddb gpurun_out/r02a/kWRITE_SIZE)" gpurun_out/r02a/pmc_traffic.json --only kalman_scan_kernel --suffix '@S=256,T=64' --into gpurun_out/r02a/pmc_traffic.json > /dev/null
python tools/pmc_traffic.py "$(finddb gpurun_out/r02a/kFETCH_SIZE)" "$(finddb gpurun_out/r02a/kWRITE_SIZE)" gpurun_out/r02a/pmc_traffic.json --only kalman_fuse_kernel --suffix '@P=78643200' --into gpurun_out/r02a/pmc_traffic.json > /dev/null
rm -rf gpurun_out/r02a/kFETCH_SIZE gpurun_out/r02a/kWRITE_SIZE
head -c 600 gpurun_out/r02a/bench.json; echo; tail -5 gpurun_out/r02a/bench.err
cat gpurun_out/r02a/c2.json | head -c 1500; echo; tail -3 gpurun_out/r02a/c2.err
cat gpurun_out/r02a/c5.json | head -c 1500; echo; tail -3 gpurun_out/r02a/c5.err
