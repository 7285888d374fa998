# manual helper: the headline workload with several builds of the library.  usage: run_lib_set.sh "" _variantA _variantB ...
for rep in 1 2; do
  for v in "$@"; do
    RLSTED_LIB=$GRAFT_REPO_ROOT/rescan_line_sted_amd/_lib/librlsted$v.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-2048 --no-accuracy --steps 30 2>/dev/null > /tmp/ab.json
    python -c "import json; d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('lib$v', round(d['value']), {k: round(v*1e3,1) for k,v in d['roofline']['kernel_avg_ms'].items() if k in ('rowpass_RATIO','rowpass_UPDATE','colconv_H')})"
  done
done
