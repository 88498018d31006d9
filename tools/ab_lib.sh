# usage: bash tools/ab_lib.sh A.so B.so [reps]  -> fps for each library, alternating, on the same box
A=$1; B=$2; R=${3:-2}
cp kfnet_amd/libkfnet_hip.so /tmp/kfn_keep.so
for i in $(seq $R); do for L in $A $B; do
  cp $L kfnet_amd/libkfnet_hip.so
  echo "$L: $(python bench.py --steps 136 --no-cpu-baseline --no-kalman-roofline --no-host-streamed --detail /tmp/ab_detail.json >/dev/null 2>&1; python -c "
import json,sys; d=json.load(open('/tmp/ab_detail.json')); k=d['per_kernel_ms_per_batch']; print(d['value'], {n.split('#')[0]:v for n,v in k.items() if v>0.5})")"
done; done
cp /tmp/kfn_keep.so kfnet_amd/libkfnet_hip.so
