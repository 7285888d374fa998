# manual helper: the round's standard GPU pass -> gpurun_out/r02/ (tests, bench lines, rocprof kernel stats, PMC passes)
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "pytest_exit=$?" >> $O/gpu_tests.log
tail -3 $O/gpu_tests.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench_exit=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-accuracy --no-2048 > $O/bench_prof.json 2> $O/bench_prof.err; echo "prof_exit=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-accuracy --no-2048 > $O/pmc_f.json 2> $O/pmc_f.err; echo "pmc_f_exit=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-accuracy --no-2048 > $O/pmc_w.json 2> $O/pmc_w.err; echo "pmc_w_exit=$?"
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_rank.json 2> $O/bench_rank.err; echo "rank_exit=$?"
timeout -k 10 600 python bench.py --size 2048 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_2048.json 2> $O/bench_2048.err; echo "b2048_exit=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2048 -- python bench.py --size 2048 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_2048_prof.json 2> $O/bench_2048_prof.err; echo "p2048_exit=$?"
echo done
