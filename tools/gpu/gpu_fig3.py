"""Manual helper (not a test): throughput of simulate_imaging at the reference's own sizes
(line_sted_figure_3.py:41-63: 128x128 test object, psf_width 25) -> profiles/r02/fig3_throughput.json."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import line_sted_figure_3 as fig3
from oracle import figure3_oracle as f3
objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
obj = objs['lines'].astype(np.float64) / 255 + 1e-6
out = {'object': 'test_object_lines 128x128', 'psf_width': 25, 'runs': []}
cases = [('descan_point', 3, 1, 25), ('nondescan_multipoint', 3, 1, 25), ('descan_line', 3, 4, int(0.45 * 128)),
         ('rescan_line', 3, 4, int(0.45 * 128)), ('descan_point', 1, 1, 25)]
for t, R, n_or, pad in cases:
    fig3.simulate_imaging(obj, t, 25, R, n_or, 1, pad)           # warm-up (allocations)
    frames = []
    t0 = time.perf_counter()
    res = fig3.simulate_imaging(obj, t, 25, R, n_or, 1, pad, generate_figure=lambda *a: frames.append(1))
    el = time.perf_counter() - t0
    t0 = time.perf_counter()
    fig3.simulate_imaging(obj, t, 25, R, n_or, 1, pad)
    el_noframes = time.perf_counter() - t0
    n_pos = len(res['scan_positions'])
    scans = n_pos * (1 + (1 if t in ('descan_point', 'nondescan_multipoint') else n_or))     # both passes
    r = {'imaging_type': t, 'R': R, 'orientations': n_or, 'pad': pad, 'scan_positions': n_pos, 'frames_rendered': len(frames),
         'seconds_with_frames': el, 'seconds_without_frames': el_noframes,
         'scan_positions_per_s': scans / el_noframes}
    out['runs'].append(r)
    print(json.dumps(r), flush=True)
# CPU oracle on the smallest case for scale
t0 = time.perf_counter()
f3.simulate_imaging(obj, 'descan_line', 25, 3, 4, 1, int(0.45 * 128))
out['cpu_oracle_descan_line_R3_4_orientations_s'] = time.perf_counter() - t0
print(out['cpu_oracle_descan_line_R3_4_orientations_s'])
os.makedirs(os.path.join(ROOT, 'gpurun_out', 'r02'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'r02', 'fig3_throughput.json'), 'w'), indent=1)
