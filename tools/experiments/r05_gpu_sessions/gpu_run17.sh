#!/bin/bash
# PNG -> .npy path: what the producer thread spends its time on (page-locking, decode, waiting for a buffer)
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python - <<'PY'
import time, torch
torch.cuda.init(); torch.zeros(1, device='cuda')
for style in ('empty(pin_memory=True)', 'empty().pin_memory()'):
    ts = []
    for i in range(4):
        t = time.perf_counter()
        a = torch.empty((32, 480, 640, 3), dtype=torch.uint8, pin_memory=True) if style.startswith('empty(pin') else torch.empty((32, 480, 640, 3), dtype=torch.uint8).pin_memory()
        ts.append((time.perf_counter() - t) * 1e3)
    print('29.5 MB page-locked by %-26s: %s ms' % (style, ' '.join('%.1f' % v for v in ts)))
PY
run() { tag=$1; shift
python bench.py --no-cpu-baseline --no-alt-modes --no-kalman-roofline --no-extra-configs --min-seconds 0.5 "$@" > gpurun_out/bench_png_$tag.json 2> gpurun_out/bench_png_$tag.err
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_png_$tag.json').read().strip().splitlines()[-1])
e=d['eval_png_end_to_end']
print('%-22s'%'$tag', d['value'], d.get('value_streamed'), e['value'], e['fraction_of_host_streamed'], e['seconds'], e['gpu_busy_pct'], e['first_chunks'], {k:v for k,v in e['consumer_thread_seconds'].items() if k!='chunks'})
PY
}
for rep in 1 2 3; do
run native2_w32_$rep
run native2_noramp_w32_$rep --eval-ramp 0
done 2>&1 | tee gpurun_out/r05_eval_png_native_ab2.log
