set -x
export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/r02c
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -W ignore -k "winograd_fused" > $O/tests_fused.log 2>&1; echo "pytest rc=$?"
tail -3 $O/tests_fused.log
timeout 600 python tools/mb_wino.py > $O/mb_wino.log 2>&1; echo "mb rc=$?"
cat $O/mb_wino.log
finddb() { find "$1" -name '*.db' | head -1; }
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  i=$((i+1))
  ( cd /tmp && MB_FUSED_ONLY=1 MB_LAYERS=conv2b,conv4b timeout 300 rocprofv3 --kernel-trace --pmc $G -d $O/sq$i -- python $R/tools/mb_wino.py > /dev/null 2> $O/sq$i.err )
done
python tools/pmc_sq.py $O/pmc_sq_wino2.json "$(finddb $O/sq1)" "$(finddb $O/sq2)" "$(finddb $O/sq3)"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02c/pmc_sq_wino2.json'))
for k,v in d.items():
    if 'wino2' in k: print(k, json.dumps(v, indent=0))
PY
rm -rf $O/sq1 $O/sq2 $O/sq3
