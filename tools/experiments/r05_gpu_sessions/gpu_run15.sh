#!/bin/bash
# PNG -> .npy path: first-chunk ramp, hipGraph replay of full tower batches, decode threads
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { tag=$1; shift
python bench.py --no-cpu-baseline --no-alt-modes --no-kalman-roofline --no-extra-configs --min-seconds 0.5 "$@" > gpurun_out/bench_png_$tag.json 2> gpurun_out/bench_png_$tag.err
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_png_$tag.json').read().strip().splitlines()[-1])
e=d['eval_png_end_to_end']
print('%-22s'%'$tag', d['value'], d.get('value_streamed'), e['value'], e['fraction_of_host_streamed'], e['seconds'], e['gpu_busy_pct'], e['first_chunks'], {k:v for k,v in e['consumer_thread_seconds'].items() if k!='chunks'})
PY
}
for rep in 1 2; do
run ramp_w32_$rep
run noramp_w32_$rep --eval-ramp 0
run ramp_graph_w32_$rep --graph
run ramp_w16_$rep --decode-workers 16
run ramp_graph_w16_$rep --graph --decode-workers 16
run ramp_w8_$rep --decode-workers 8
done 2>&1 | tee gpurun_out/r05_eval_png_ab.log
