#!/usr/bin/env python3
"""Golden vectors G11 for SURVEY.md row f-3: what the reference's `simulate_imaging`
(figure_generation/line_sted_figure_3.py:76-273) hands to its figure code.

That module must never be imported: it calls main() at module level (:411), hours of figure generation.
This script parses the file, takes the FUNCTION DEFINITIONS simulate_imaging / rotate / shift / scale_y
out of the syntax tree and executes only those, in a namespace whose `generate_figure` records its
arguments and whose `animate` does nothing.  Nothing of the reference is copied into the repository:
the file stored is inputs + recorded outputs.

Run in the build container only:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 -B /root/repo/tests/golden/make_golden_fig3.py
"""
import ast
import contextlib
import io
import json
import os
import sys
import warnings

import numpy as np
import scipy

REF_FILE = '/root/reference/figure_generation/line_sted_figure_3.py'
HERE = os.path.dirname(os.path.abspath(__file__))
WANTED = ('simulate_imaging', 'rotate', 'shift', 'scale_y')
warnings.filterwarnings('ignore')

META = json.dumps({'numpy': np.__version__, 'scipy': scipy.__version__, 'python': sys.version.split()[0],
                   'generator': 'tests/golden/make_golden_fig3.py',
                   'source': 'function definitions %s of figure_generation/line_sted_figure_3.py, main() not run' % (WANTED,)})


def reference_functions(record):
    tree = ast.parse(open(REF_FILE).read(), REF_FILE)
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert sorted(d.name for d in defs) == sorted(WANTED)
    from scipy.ndimage import gaussian_filter, interpolation
    ns = {'np': np, 'os': os, 'warnings': warnings, 'gaussian_filter': gaussian_filter, 'interpolation': interpolation,
          'generate_figure': record, 'animate': lambda *a, **k: None}
    exec(compile(ast.Module(body=defs, type_ignores=[]), REF_FILE, 'exec'), ns)
    return ns['simulate_imaging']


def main():
    objs = np.load(os.path.join(HERE, 'objects.npz'))
    # 32 x 32 objects: the shipped 128 x 128 'rings' and 'lines' decimated by 4, scaled as main() does (:44)
    objects = {name: objs[name][:, ::4, ::4].astype(np.float64) / 255 + 1e-6 for name in ('rings', 'lines')}
    out = {'meta': np.array(META)}
    for name, o in objects.items():
        out['obj/' + name] = o
    psf_width = 8
    cases = []
    for R in (1, 2):
        cases += [('rings', 'descan_point', R, 1, psf_width), ('rings', 'nondescan_multipoint', R, 1, psf_width),
                  ('lines', 'descan_line', R, 2, int(0.45 * 32)), ('lines', 'rescan_line', R, 2, int(0.45 * 32))]
    cases.append(('rings', 'rescan_line', 3, 4, int(0.45 * 32)))
    names = []
    for obj_name, imaging_type, R, n_orient, pad in cases:
        frames = []

        def record(filename, obj, excitation, glow, inst, cum, new_signal, reconstruction, pulses, exposures):
            frames.append((os.path.basename(filename), [np.array(a, dtype=np.float64) for a in
                           (excitation, glow, inst, cum, new_signal, reconstruction)], pulses, exposures))
        simulate = reference_functions(record)
        with contextlib.redirect_stdout(io.StringIO()):
            simulate(objects[obj_name], imaging_type, psf_width, R, n_orient, pulses_per_position=1, pad=pad,
                     comparison_name='case')
        key = '%s_%s_R%d' % (obj_name, imaging_type, R)
        names.append(key)
        # the file name carries the orientation and the scan position of the frame
        rots = np.array([int(f[0].split('deg_')[0].rsplit('_', 1)[1]) for f in frames])
        pos = np.array([int(f[0].rsplit('_', 1)[1].split('.')[0]) for f in frames])
        out[key + '/args'] = np.array([psf_width, R, n_orient, 1, pad], dtype=np.float64)
        out[key + '/frame_rot'] = rots
        out[key + '/frame_pos'] = pos
        out[key + '/frame_pulses'] = np.array([f[2] for f in frames], dtype=np.float64)
        out[key + '/frame_exposures'] = np.array([-1 if f[3] == 'N/A' else f[3] for f in frames], dtype=np.float64)
        # every frame: the sum of each of the six arrays; a few frames in full
        out[key + '/frame_sums'] = np.array([[a.sum() for a in f[1]] for f in frames])
        out[key + '/frame_maxs'] = np.array([[a.max() for a in f[1]] for f in frames])
        last_of_rot = [np.nonzero(rots == r)[0][-1] for r in sorted(set(rots.tolist()))]
        full = sorted(set(last_of_rot + [len(frames) // 3]))
        out[key + '/full_index'] = np.array(full)
        out[key + '/full'] = np.array([frames[i][1] for i in full])      # (n_full, 6, n_y, n_x)
        print(key, 'frames', len(frames), 'full', full)
    out['cases'] = np.array(names)
    path = os.path.join(HERE, 'g11_fig3.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
