#!/bin/bash
# manual helper: the round's record -- profile set, the default bench line, the sweep bench line -> gpurun_out/rNN/    usage: tools/gpu/run_round.sh r04
R=${1:-r04}
O=gpurun_out/$R
mkdir -p $O
tools/gpu/profile_round.sh $R > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
timeout -k 10 600 python3 bench.py > $O/bench_$R.json 2> $O/bench_$R.err; echo "bench rc $?"; tail -2 $O/bench_$R.err
timeout -k 10 300 python3 bench.py --workload fig2sweep --no-extra-legs --no-cpu-baseline --no-accuracy --steps 10 > $O/bench_fig2sweep_$R.json 2> /dev/null; echo "sweep rc $?"
# the N-rank code path at world size 1 (RCCL communicator of the C ABI: barrier, max, broadcast of the PSF sets, device gather)
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 timeout -k 10 300 python3 bench.py --gpus 1 --workload fig2sweep --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --no-accuracy > $O/bench_launcher_world1.json 2> $O/bench_launcher_world1.err; echo "world-1 launcher path rc $?"
python3 - <<PY
import json
d = json.load(open('$O/bench_$R.json'))
print(d['value'], d['roofline']['frac'], d['roofline']['whole_path'], d.get('accuracy', {}).get('normwise'), d.get('accuracy', {}).get('pixelwise'))
for k in ('size_2048', 'line_rescan_512', 'f64_512', 'point_2048', 'f64_2048', 'size_4096_k100'):
    print(k, d[k].get('value'), d[k].get('whole_path', {}).get('frac'), d[k].get('rl_iteration_traffic'), d[k].get('error'))
print(d.get('cpu_baseline'))
s = json.load(open('$O/bench_fig2sweep_$R.json'))['fig2_sweep']
print(s['frames_per_s'], s['seconds_run_max_over_ranks'], s['seconds_first_pass_with_plan_setup'])
w = json.load(open('$O/bench_launcher_world1.json'))
print('world 1 through the communicator:', w['value'], w['final_gather'], w['fig2_sweep']['psf_sets'], w['fig2_sweep']['gather_transport'], w['fig2_sweep']['gather_ms'])
PY
