# manual helper: A/B of one environment knob on the headline workload.  usage: run_env_ab.sh NAME VALUE_A VALUE_B [reps]
for rep in $(seq 1 ${4:-2}); do
  for v in "$2" "$3"; do
    env "$1=$v" timeout -k 10 200 python bench.py --no-cpu-baseline --no-2048 --no-accuracy --steps 30 2>/dev/null > /tmp/ab.json
    python -c "import json; d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('$1=$v', round(d['value']), round(d['roofline']['kernels_in_flight'], 2))"
  done
done
