#!/bin/bash
# The round's profile set: rocprofv3 kernel-trace statistics and the two PMC passes (FETCH_SIZE / WRITE_SIZE, each in a run of
# its own) of bench.py at 512 x 512 (headline) and 2048 x 2048 (config 3), the same two PMC passes for the bench line's other legs
# (one short cycle each: tools/gpu/prof_cycle.py), and two passes of SQ counters over one cycle of the headline shape, all
# summarised into gpurun_out/rNN/prof -- copy what is to be judged into profiles/rNN/.
#     usage (on the GPU box, from the repo root):  tools/gpu/profile_round.sh r04
set -uo pipefail
R=${1:-r04}
OUT=gpurun_out/$R/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS512="--steps 2 --warmup 1 --batch 256 --no-cpu-baseline --no-extra-legs --no-accuracy"
ARGS2048="--size 2048 --steps 1 --warmup 1 --no-cpu-baseline"
run() { echo "== $*" >&2; timeout -k 10 600 "$@"; }
f() { find $OUT/$1 -name "*$2" | head -1; }
run rocprofv3 --kernel-trace --stats -d $OUT/stats512 --output-format csv -- python3 bench.py $ARGS512 > $OUT/bench512_under_rocprof.json 2> $OUT/stats512.err
run rocprofv3 --kernel-trace --stats -d $OUT/stats2048 --output-format csv -- python3 bench.py $ARGS2048 > $OUT/bench2048_under_rocprof.json 2> $OUT/stats2048.err
for c in FETCH_SIZE WRITE_SIZE; do
  run rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc512_$c --output-format csv -- python3 bench.py $ARGS512 > /dev/null 2> $OUT/pmc512_$c.err
  run rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc2048_$c --output-format csv -- python3 bench.py $ARGS2048 > /dev/null 2> $OUT/pmc2048_$c.err
done
python3 tools/pmc_traffic.py "$(f pmc512_FETCH_SIZE counter_collection.csv)" "$(f pmc512_WRITE_SIZE counter_collection.csv)" $OUT/pmc_traffic.json 32 512 1
python3 tools/pmc_traffic.py "$(f pmc2048_FETCH_SIZE counter_collection.csv)" "$(f pmc2048_WRITE_SIZE counter_collection.csv)" $OUT/pmc_traffic_2048.json 6 2048 4
cp "$(f stats512 kernel_stats.csv)" $OUT/bench_kernel_stats.csv
cp "$(f stats2048 kernel_stats.csv)" $OUT/bench_2048_kernel_stats.csv
python3 tools/pmc_summary.py "$(f pmc512_FETCH_SIZE counter_collection.csv)" "$(f pmc512_WRITE_SIZE counter_collection.csv)" > $OUT/summary_pmc512.txt
python3 tools/pmc_summary.py "$(f pmc2048_FETCH_SIZE counter_collection.csv)" "$(f pmc2048_WRITE_SIZE counter_collection.csv)" > $OUT/summary_pmc2048.txt
cp "$(f pmc512_FETCH_SIZE counter_collection.csv)" $OUT/pmc_fetch_size.csv; cp "$(f pmc512_WRITE_SIZE counter_collection.csv)" $OUT/pmc_write_size.csv
rm -rf $OUT/pmc512_* $OUT/pmc2048_F* $OUT/pmc2048_W* $OUT/stats512 $OUT/stats2048 2>/dev/null || true
# the other legs of the bench line: SIZE VIEWS BATCH K DTYPE (one short cycle each)
for leg in "512 4 256 3 f32" "512 1 256 3 f64" "2048 1 32 3 f32" "2048 1 16 3 f64" "4096 1 8 3 f32"; do
  set -- $leg
  tag=$1_$2v_$5
  for c in FETCH_SIZE WRITE_SIZE; do
    run rocprofv3 --kernel-trace --pmc $c -d $OUT/leg_${tag}_$c --output-format csv -- python3 tools/gpu/prof_cycle.py $1 $2 $3 $4 0 $5 $OUT/fl_$tag.txt > /dev/null 2> $OUT/leg_${tag}_$c.err
  done
  python3 tools/pmc_traffic.py "$(f leg_${tag}_FETCH_SIZE counter_collection.csv)" "$(f leg_${tag}_WRITE_SIZE counter_collection.csv)" $OUT/pmc_traffic_$tag.json "$(cat $OUT/fl_$tag.txt)" $1 $2 $5 > /dev/null
  rm -rf $OUT/leg_${tag}_* $OUT/fl_$tag.txt
done
# SQ counters of the headline kernels, one lane (kernels alone), K = 3: two passes of 8
SQA="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
SQB="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run rocprofv3 --kernel-trace --pmc $SQA -d $OUT/sqa --output-format csv -- python3 tools/gpu/prof_cycle.py 512 1 64 3 1 > /dev/null 2> $OUT/sqa.err
run rocprofv3 --kernel-trace --pmc $SQB -d $OUT/sqb --output-format csv -- python3 tools/gpu/prof_cycle.py 512 1 64 3 1 > /dev/null 2> $OUT/sqb.err
{ echo "# rocprofv3 --kernel-trace --pmc <8 SQ counters> -- python3 tools/gpu/prof_cycle.py 512 1 64 3 1 (two passes; per-kernel averages over the launches of ONE cycle, K = 3, one lane; tools/pmc_summary.py).  Quad-cycle units for the *_CYCLES / WAIT / ACTIVE counters."; python3 tools/pmc_summary.py "$(f sqa counter_collection.csv)" "$(f sqb counter_collection.csv)"; } > $OUT/sq_counters_512.txt
rm -rf $OUT/sqa $OUT/sqb
ls -la $OUT
