#!/bin/bash
# PMC comparison of two wino3_prof binaries with byte-identical loops: tools/mb/pmc_pair.sh binA binB [args...]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A=$1; B=$2; shift 2
ARGS="${@:-16 120 160 512 512}"
if [ -n "$PMC_SETS_FILE" ]; then mapfile -t SETS < $R/$PMC_SETS_FILE; else
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
      "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
      "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
      "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum")
fi
for bin in $A $B; do
  i=0
  for s in "${SETS[@]}"; do
    timeout 100 rocprofv3 --pmc $s --kernel-trace -d /tmp/pmc_${bin}_$i -o p --output-format csv -- $R/tools/mb/$bin $ARGS > /dev/null 2>&1
    f=$(find /tmp/pmc_${bin}_$i -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$bin" <<'PY'
import csv, sys, collections
f, b = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(f)):
        if 'wino3' in r.get('Kernel_Name', ''):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
except Exception as e:
    print(b, 'no data', e)
for k, v in acc.items():
    print('%-12s %-44s %16.0f' % (b, k, sum(v) / len(v)))
PY
    i=$((i+1))
  done
done
