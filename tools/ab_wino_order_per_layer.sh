export TMPDIR=/tmp
R=$(pwd)
mkdir -p gpurun_out/order
for L in conv2b conv3b conv6 conv5; do
for o in 0 1; do
  ( cd /tmp && MB_LAYERS=$L KFN_WINO_ORDER=$o MB_FUSED_ONLY=1 MB_BATCH=16 timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/order/f -- python $R/tools/mb_wino.py 2>/dev/null | grep -v amdgpu | cut -c1-62 )
  python - "$(find gpurun_out/order/f -name '*.db' | head -1)" $L $o <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for name, n, avg in c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name"):
    if 'wino' in name:
        print('   %s order %s: FETCH_SIZE*2 avg %.3f GB over %d launches' % (sys.argv[2], sys.argv[3], avg * 2 * 1024 / 1e9, n))
PY
  rm -rf gpurun_out/order/f
done; done
