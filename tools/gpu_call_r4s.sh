cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4s
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -q -x -k "oflow or small_sequence or unfused" 2>&1 | tail -3
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/r4s/sq -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --batch 32 --min-seconds 0 --no-cpu-baseline --no-host-streamed --no-alt-modes --no-kalman-roofline --no-config3 --no-extra-configs --no-eval-png > $GRAFT_REPO_ROOT/gpurun_out/r4s/bench_pmc.json 2> /dev/null
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, glob
db=glob.glob('gpurun_out/r4s/sq/**/*.db', recursive=True)[0]
c=sqlite3.connect(db)
for name,cn,n,avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%oflow%' group by kernel_name, counter_name"):
    print(name[:60], cn, n, avg)
PY
python -c "
import json; d=json.loads(open('gpurun_out/r4s/bench_pmc.json').read().strip().splitlines()[-1]); print({k:v for k,v in d['per_kernel_ms_per_batch'].items() if 'oflow' in k})"
rm -rf gpurun_out/r4s/sq
