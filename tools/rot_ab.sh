for m in 0 1 2; do
  echo "ROT=$m"
  KFN_CONV_ROT=$m python bench.py --steps 68 --no-cpu-baseline --no-kalman-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value']); print({k:(v['ms'], v['executed_tflops']) for k,v in d['kernels_ms_per_batch'].items() if 'conv_mfma_kernel<5' in k or 'conv_mfma_kernel<3, 1, 2, 2, 32, 0' in k})"
done
