export TMPDIR=/tmp
R=$(pwd)
mkdir -p gpurun_out/order
for o in 1 2; do   # kfn_conv_desc.wino_order: 1 = tile blocks fastest, 2 = channel groups fastest
  echo "== wino_order=$o"
  MB_WINO_ORDER=$o MB_FUSED_ONLY=1 MB_BATCH=16 python tools/mb_wino.py 2>&1 | grep -v amdgpu.ids | cut -c1-70
  MB_WINO_ORDER=$o python tools/mb_s2.py 2>&1 | grep -v amdgpu.ids | cut -c1-90
  ( cd /tmp && MB_WINO_ORDER=$o MB_FUSED_ONLY=1 MB_BATCH=16 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/order/f$o -- python $R/tools/mb_wino.py > /dev/null 2>&1 )
  ( cd /tmp && MB_WINO_ORDER=$o timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/order/s$o -- python $R/tools/mb_s2.py > /dev/null 2>&1 )
  for k in f s; do
    python - "$(find gpurun_out/order/$k$o -name '*.db' | head -1)" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for name, n, avg in c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name"):
    if 'wino' in name:
        import re
        print('   FETCH_SIZE*2  %-22s launches %3d  avg %.3f GB' % (re.search(r'(wino\w*_kernel<?\w*>?)', name).group(1), n, avg * 2 * 1024 / 1e9))
PY
  done
done
rm -rf gpurun_out/order
