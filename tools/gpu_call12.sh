export TMPDIR=/tmp
mkdir -p gpurun_out/r02c
timeout 600 python bench.py --no-cpu-baseline --no-host-streamed --no-alt-modes --no-kalman-roofline > gpurun_out/r02c/bench.json 2> gpurun_out/r02c/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r02c/bench.json'))
print('fps', d['value'], 'ms/step', d['ms_per_step'])
pk = d['per_kernel_ms_per_batch']
print('heavy ms', round(sum(pk.values()), 3))
for k, v in pk.items():
    if v > 0.15 or 'tail' in k or 'flow' in k or 'conv6' in k:
        print('  %-46s %.3f' % (k, v))
PY
