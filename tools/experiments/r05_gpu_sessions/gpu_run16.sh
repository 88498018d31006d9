#!/bin/bash
# PNG -> .npy path: the library's native decoder (kfn_decode_png_rgb8) against PIL on a thread pool
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
run native_w32_$rep
KFN_PNG_DECODER=pil run pil_w32_$rep
run native_w16_$rep --decode-workers 16
run native_w64_$rep --decode-workers 64
run native_noramp_w32_$rep --eval-ramp 0
KFN_PNG_DECODER=pil run pil_w16_$rep --decode-workers 16
done 2>&1 | tee gpurun_out/r05_eval_png_native_ab.log
timeout 300 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_weights.py -q -x -k "png or eval or model_folder or streamed" -p no:cacheprovider 2>&1 | tail -3
