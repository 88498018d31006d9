cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4t
for b in 4 6 8 10 12 16 24 32; do
  python bench.py --steps 96 --batch $b --min-seconds 1.5 --no-cpu-baseline --no-host-streamed --no-alt-modes --no-kalman-roofline --no-config3 --no-extra-configs --no-eval-png 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels_ms_per_batch']
print('batch', d['config']['tower_batch'], 'fps', d['value'], 'wino4 exec TF', k['wino4_kernel']['executed_tflops'], 'wino_s2', k['wino_s2_kernel']['executed_tflops'])" >> gpurun_out/r4t/batch_sweep.log
done
cat gpurun_out/r4t/batch_sweep.log
