export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/r02e
mkdir -p $O
finddb() { find "$1" -name '*.db' | head -1; }
for L in conv1b conv2b conv4b; do
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  ( cd /tmp && MB_FUSED_ONLY=1 MB_LAYERS=$L timeout 300 rocprofv3 --kernel-trace --pmc $G -d $O/sq$i -- python $R/tools/mb_wino.py > /dev/null 2> $O/sq$i.err )
done
python tools/pmc_sq.py $O/pmc_$L.json "$(finddb $O/sq1)" "$(finddb $O/sq2)" "$(finddb $O/sq3)"
rm -rf $O/sq1 $O/sq2 $O/sq3
python - <<PY
import json
d=json.load(open('$O/pmc_$L.json'))
v=d['wino2_kernel']
w=v['SQ_WAVE_CYCLES']
print('$L', 'MfmaUtil %.1f%%'%v['MfmaUtil_pct'], 'WAIT_ANY %.1f%%'%(100*v['SQ_WAIT_ANY']/w), 'WAIT_INST %.1f%%'%(100*v['SQ_WAIT_INST_ANY']/w), 'ACTIVE %.1f%%'%(100*v['SQ_ACTIVE_INST_ANY']/w),
      'VALU %.3g SALU %.3g VMEM %.3g LDS %.3g'%(v['SQ_INSTS_VALU'], v['SQ_INSTS_SALU'], v['SQ_INSTS_VMEM_RD'], v['SQ_INSTS_LDS']), 'mfma_cycles %.4g'%v['SQ_VALU_MFMA_BUSY_CYCLES'])
PY
done
