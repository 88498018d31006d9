#!/bin/bash
# round-5 GPU call 7: the real-data block with three chunks in flight, decode threads 8 / 16 / 32; the streaming tests
mkdir -p gpurun_out && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider -k "streamed or eval" > gpurun_out/gpu_tests7.log 2>&1; tail -2 gpurun_out/gpu_tests7.log
for WK in 8 16 32; do
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-alt-modes --no-kalman-roofline --decode-workers $WK > gpurun_out/bench_eval_workers$WK.json 2> gpurun_out/bench_eval_workers$WK.err
  python -c "import json;d=json.load(open('gpurun_out/bench_eval_workers$WK.json'));e=d['eval_png_end_to_end'];print($WK,d['value'],d['value_streamed'],e['value'],e['fraction_of_host_streamed'],e['gpu_busy_pct'])"
done
