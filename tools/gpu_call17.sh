export TMPDIR=/tmp
for b in 16 32 24; do
timeout 600 python bench.py --batch $b --no-cpu-baseline --no-host-streamed --no-alt-modes --no-kalman-roofline > gpurun_out/b$b.json 2> gpurun_out/b$b.err; echo "batch $b rc=$?"
python - $b <<'PY'
import json, sys
d = json.load(open('gpurun_out/b%s.json' % sys.argv[1]))
print('  fps', d['value'], 'ms/step', d['ms_per_step'], 'tower_batch', d['config'].get('tower_batch'))
PY
done
