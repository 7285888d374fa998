#!/bin/bash
# Manual helper: kernel trace + FETCH/WRITE + two SQ counter passes over ONE short cycle of a multi-view plan on the long
# transforms (one lane, kernels alone).   usage (GPU box, repo root): tools/gpu/prof_split.sh SIZE VIEWS BATCH OUTDIR
set -euo pipefail
N=${1:-2048}; V=${2:-4}; B=${3:-6}; OUT=${4:-gpurun_out/r03/split}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { echo "== $*" >&2; timeout -k 10 300 "$@"; }
run rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 tools/gpu/prof_cycle.py $N $V $B 3 1 > $OUT/trace.log 2>&1
run rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- python3 tools/gpu/prof_cycle.py $N $V $B 3 1 > $OUT/fetch.log 2>&1
run rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write --output-format csv -- python3 tools/gpu/prof_cycle.py $N $V $B 3 1 > $OUT/write.log 2>&1
run rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES -d $OUT/sq1 --output-format csv -- python3 tools/gpu/prof_cycle.py $N $V $B 3 1 > $OUT/sq1.log 2>&1
run rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS -d $OUT/sq2 --output-format csv -- python3 tools/gpu/prof_cycle.py $N $V $B 3 1 > $OUT/sq2.log 2>&1
f() { find $OUT/$1 -name "*$2" | head -1; }
cp "$(f trace kernel_stats.csv)" $OUT/kernel_stats.csv
python3 tools/pmc_summary.py "$(f fetch counter_collection.csv)" "$(f write counter_collection.csv)" > $OUT/summary_traffic.txt
python3 tools/pmc_summary.py "$(f sq1 counter_collection.csv)" "$(f sq2 counter_collection.csv)" > $OUT/summary_sq.txt
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/sq1 $OUT/sq2
