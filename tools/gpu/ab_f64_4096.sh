#!/bin/bash
# manual helper: float64 at 4096^2 (L = 4608) -- parity tests on that path, then an A/B of library builds.   usage: tools/gpu/ab_f64_4096.sh OUTDIR LIB...
O=$1; shift
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -k "4096 or 4000 or tile_runs or ragged or edge" > $O/gpu_tests_4608.log 2>&1; echo "pytest rc $?"; tail -3 $O/gpu_tests_4608.log
python3 tools/gpu/ab_bench.py --size 4096 --dtype f64 --batch 4 --rounds 3 --reps 2 "$@" > $O/ab_f64_4096.log 2>&1; cut -c1-170 $O/ab_f64_4096.log
python3 tools/gpu/ab_bench.py --size 4096 --dtype f64 --views 2 --batch 2 --rounds 2 --reps 2 "$@" > $O/ab_f64_4096v2.log 2>&1; cut -c1-170 $O/ab_f64_4096v2.log
