set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r02c
timeout 1200 python -m pytest tests -m gpu -q -W ignore -x > gpurun_out/r02c/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02c/tests.log
tail -6 gpurun_out/r02c/tests.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02c/bench.json 2> gpurun_out/r02c/bench.err; echo "bench rc=$?"
head -c 700 gpurun_out/r02c/bench.json; echo; tail -3 gpurun_out/r02c/bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r02c/bench.json'))
print(json.dumps(d['roofline'], indent=0)[:900])
for k, v in list(d['kernels_ms_per_batch'].items())[:8]:
    print(k, v)
PY
