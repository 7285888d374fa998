# manual helper: the headline workload under several environment settings.  usage: run_env_set.sh "A=1 B=2" "A=3" ...
for rep in 1 2; do
  for setting in "$@"; do
    env $setting timeout -k 10 200 python bench.py --no-cpu-baseline --no-2048 --no-accuracy --steps 30 2>/dev/null > /tmp/ab.json
    python -c "import json; d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('$setting', round(d['value']), round(d['roofline']['kernels_in_flight'], 2), d['roofline']['frames_per_launch'])"
  done
done
