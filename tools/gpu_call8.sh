export TMPDIR=/tmp
O=gpurun_out/r02f
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -W ignore -x > $O/tests.log 2>&1; echo "pytest rc=$?"
tail -4 $O/tests.log
timeout 600 python bench.py --no-cpu-baseline --no-alt-modes --no-host-streamed > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02f/bench.json'))
print(d['value'], d['ms_per_step'], d['repetitions'])
for k,v in d['per_kernel_ms_per_batch'].items(): print('%-40s %8.3f'%(k,v))
for k,v in d['kernels_ms_per_batch'].items(): print(k, v)
PY
