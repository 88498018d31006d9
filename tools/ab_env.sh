# usage: bash tools/ab_env.sh VAR v1 v2 ...   -> fps + main conv kernels for each value
VAR=$1; shift
for m in "$@"; do
  echo "$VAR=$m"
  env $VAR=$m python bench.py --steps 68 --no-cpu-baseline --no-kalman-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value']); print({k[17:]:(v['ms'], v['executed_tflops']) for k,v in d['kernels_ms_per_batch'].items() if 'conv_mfma_kernel' in k and v['ms']>1}); print('wino_out', d['kernels_ms_per_batch'].get('wino_output_kernel',{}).get('ms'))"
done
