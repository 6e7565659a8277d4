#!/bin/bash
# PMC passes (separate runs, <= 4 counters each) over tools/micro_one.py; prints per-kernel counter averages for kernels matching $1.
# usage: tools/pmc_one.sh <kernel-substring> <micro_one args...>
pat=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_VMEM_TA_ADDR_FIFO_FULL" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$i -- python $root/tools/micro_one.py "$@" > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name '*counter_collection.csv' | head -1)
  [ -z "$f" ] && { echo "pass $i: no counter file"; tail -3 /tmp/pmc_$i.log; continue; }
  python - "$f" "$pat" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Kernel_Name']:
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in acc: print('%-32s %14.4g per launch (%d launches)' % (k, acc[k] / n[k], n[k]))
PY
done
