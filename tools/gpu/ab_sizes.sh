#!/bin/bash
# manual helper: the full GPU test suite, then A/B of library builds at the bench shapes.   usage: tools/gpu/ab_sizes.sh OUTDIR LIB_A LIB_B ...
O=$1; shift
mkdir -p $O
python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -4 $O/gpu_tests.log
python3 tools/gpu/ab_bench.py --rounds 5 "$@" > $O/ab_512.log 2>&1; cut -c1-150 $O/ab_512.log
python3 tools/gpu/ab_bench.py --size 512 --views 4 --batch 256 --rounds 3 --reps 2 "$@" > $O/ab_512v4.log 2>&1; cut -c1-150 $O/ab_512v4.log
python3 tools/gpu/ab_bench.py --size 2048 --batch 32 --rounds 3 --reps 2 "$@" > $O/ab_2048.log 2>&1; cut -c1-150 $O/ab_2048.log
python3 tools/gpu/ab_bench.py --size 2048 --views 4 --batch 32 --rounds 3 --reps 2 "$@" > $O/ab_2048v4.log 2>&1; cut -c1-150 $O/ab_2048v4.log
python3 tools/gpu/ab_bench.py --size 4096 --batch 8 --rounds 3 --reps 2 "$@" > $O/ab_4096.log 2>&1; cut -c1-150 $O/ab_4096.log
python3 tools/gpu/ab_bench.py --size 128 --batch 2048 --rounds 3 --reps 3 "$@" > $O/ab_128.log 2>&1; cut -c1-150 $O/ab_128.log
