#!/bin/bash
# manual helper: SQ counter passes over one short cycle of a plan (tools/gpu/prof_cycle.py), summed per kernel.
#   usage: tools/gpu/pmc_cycle.sh OUTDIR SIZE VIEWS BATCH K DTYPE
O=$(realpath -m $1); shift
mkdir -p $O
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i --output-format csv -- python3 $R/tools/gpu/prof_cycle.py "$@" > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/p$i.log; exit 1; }
done
python3 - $O <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(sys.argv[1] + '/p*/*/*counter_collection.csv'):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:90]
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:8]:
    print(k)
    print('   ', '  '.join('%s=%.4g' % (n, v) for n, v in sorted(c.items())))
PY
