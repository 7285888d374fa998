#!/bin/bash
# The round's profile set: rocprofv3 kernel-trace statistics and the two PMC passes (FETCH_SIZE / WRITE_SIZE, each in a run of
# its own) of bench.py at 512 x 512 (headline) and 2048 x 2048 (config 3), summarised into gpurun_out/rNN/ -- copy what is to be
# judged into profiles/rNN/.      usage (on the GPU box, from the repo root):  tools/gpu/profile_round.sh r03
set -euo pipefail
R=${1:-r03}
OUT=gpurun_out/$R/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS512="--steps 2 --warmup 1 --batch 256 --no-cpu-baseline --no-extra-legs --no-accuracy"
ARGS2048="--size 2048 --steps 1 --warmup 1 --no-cpu-baseline"
run() { echo "== $*" >&2; timeout -k 10 600 "$@"; }
run rocprofv3 --kernel-trace --stats -d $OUT/stats512 --output-format csv -- python3 bench.py $ARGS512 > $OUT/bench512_under_rocprof.json 2> $OUT/stats512.err
run rocprofv3 --kernel-trace --stats -d $OUT/stats2048 --output-format csv -- python3 bench.py $ARGS2048 > $OUT/bench2048_under_rocprof.json 2> $OUT/stats2048.err
for c in FETCH_SIZE WRITE_SIZE; do
  run rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc512_$c --output-format csv -- python3 bench.py $ARGS512 > /dev/null 2> $OUT/pmc512_$c.err
  run rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc2048_$c --output-format csv -- python3 bench.py $ARGS2048 > /dev/null 2> $OUT/pmc2048_$c.err
done
f() { find $OUT/$1 -name "*$2" | head -1; }
python3 tools/pmc_traffic.py "$(f pmc512_FETCH_SIZE counter_collection.csv)" "$(f pmc512_WRITE_SIZE counter_collection.csv)" $OUT/pmc_traffic.json 32 512 1
python3 tools/pmc_traffic.py "$(f pmc2048_FETCH_SIZE counter_collection.csv)" "$(f pmc2048_WRITE_SIZE counter_collection.csv)" $OUT/pmc_traffic_2048.json 6 2048 4
cp "$(f stats512 kernel_stats.csv)" $OUT/bench_kernel_stats.csv
cp "$(f stats2048 kernel_stats.csv)" $OUT/bench_2048_kernel_stats.csv
# the raw counter CSVs are large: keep the per-kernel averages only
python3 tools/pmc_summary.py "$(f pmc512_FETCH_SIZE counter_collection.csv)" "$(f pmc512_WRITE_SIZE counter_collection.csv)" > $OUT/summary_pmc512.txt
python3 tools/pmc_summary.py "$(f pmc2048_FETCH_SIZE counter_collection.csv)" "$(f pmc2048_WRITE_SIZE counter_collection.csv)" > $OUT/summary_pmc2048.txt
cp "$(f pmc512_FETCH_SIZE counter_collection.csv)" $OUT/pmc_fetch_size.csv; cp "$(f pmc512_WRITE_SIZE counter_collection.csv)" $OUT/pmc_write_size.csv
rm -rf $OUT/pmc512_* $OUT/pmc2048_F* $OUT/pmc2048_W* $OUT/stats512 $OUT/stats2048 2>/dev/null || true
ls -la $OUT
