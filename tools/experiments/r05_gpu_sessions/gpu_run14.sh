#!/bin/bash
# where the PNG -> .npy path loses its 7-10 % against the streamed path: the consumer thread's wall-time breakdown
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in 0 16; do
python bench.py --no-cpu-baseline --no-alt-modes --no-kalman-roofline --no-extra-configs --min-seconds 0.5 --decode-workers $w > gpurun_out/bench_png_w$w.json 2> gpurun_out/bench_png_w$w.err
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_png_w$w.json').read().strip().splitlines()[-1])
e=d['eval_png_end_to_end']
print('workers $w:', d['value'], d.get('value_streamed'), e['value'], e['fraction_of_host_streamed'], e['seconds'], e['gpu_busy_pct'], e['consumer_thread_seconds'])
PY
done
