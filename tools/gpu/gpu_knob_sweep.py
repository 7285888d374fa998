"""Manual helper (not a test): headline frames/s over environment knobs of the plan.
usage: gpu_knob_sweep.py "CHUNK_MB=54,LANES=2,COL_ORDER=1" "CHUNK_MB=27,LANES=4" ...   (RLSTED_ prefix implied)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from rescan_line_sted_amd import _lib
obj, psf, brightness, _ = bench.workload(512)
B = 256
rows = []
for spec in sys.argv[1:]:
    for k in [k for k in os.environ if k.startswith('RLSTED_') and k != 'RLSTED_LIB']:
        del os.environ[k]
    for kv in filter(None, spec.split(',')):
        k, v = kv.split('=')
        os.environ['RLSTED_' + k] = v
    plan = _lib.DeconvPlan(psf, B, 512, 512, dtype='f32')
    plan.set_object(np.broadcast_to(obj, (B, 512, 512)), brightness)
    plan.bench_cycles(20, 2, seed=1)
    t0 = time.perf_counter(); plan.bench_cycles(20, 8, seed=2); el = time.perf_counter() - t0
    kt, fpl = plan.time_cycle(20, seed=3)
    row = {'knobs': spec, 'frames_per_s': 8 * B / el, 'frames_per_launch': fpl, 'avg_us': {k: round(v[0] * 1e3, 1) for k, v in kt.items()}}
    rows.append(row)
    print(json.dumps(row), flush=True)
    del plan
json.dump(rows, open(os.path.join(ROOT, 'gpurun_out', 'r02', 'knobs_%s.json' % os.environ.get('TAG', 'x')), 'w'), indent=1)
