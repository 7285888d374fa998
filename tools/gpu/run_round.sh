# manual helper: the round's standard GPU pass (tests, bench line, rocprof kernel stats, world-1 launcher run)
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests.log 2>&1; echo "pytest_exit=$?" >> gpurun_out/r02/gpu_tests.log
tail -5 gpurun_out/r02/gpu_tests.log
timeout -k 10 300 python bench.py > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.err; echo "bench_exit=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/prof -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-accuracy > gpurun_out/r02/bench_prof.json 2> gpurun_out/r02/bench_prof.err; echo "prof_exit=$?"
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r02/bench_rank.json 2> gpurun_out/r02/bench_rank.err; echo "rank_exit=$?"
echo done
