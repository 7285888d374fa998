#!/usr/bin/env python3
"""Generate the golden vectors in this directory by IMPORTING THE REFERENCE.

Run in the build container only (the reference never travels to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        python3 -B /root/repo/tests/golden/make_golden.py [g1 g3 g4 g5 g8 ...]

Nothing from the reference is copied: only inputs and the outputs it computes
are stored (npz).  Versions used are recorded in every file's `meta` entry.
"""
import json
import os
import sys
import time
import warnings

import numpy as np
import scipy

REF = '/root/reference/figure_generation'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
warnings.filterwarnings('ignore')
os.environ.setdefault('MPLBACKEND', 'Agg')

import line_sted_tools as st          # noqa: E402  (the reference)
import np_tif                          # noqa: E402  (the reference)

META = json.dumps({'numpy': np.__version__, 'scipy': scipy.__version__,
                   'python': sys.version.split()[0],
                   'generator': 'tests/golden/make_golden.py'})

STEPS = (4, 6, 8, 12, 25)
BRIGHT = ((0.25, 0.0), (0.25, 9.0), (1.0, 3.0), (4.0, 27.0))


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, meta=np.array(META), **arrays)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024))


def g1_psf_report():
    """psf_report outputs (ref line_sted_tools.py:75-166) + get_width rows."""
    out = {}
    cases = []
    for psf_type in ('point', 'line'):
        for steps in STEPS:
            for exc, dep in BRIGHT:
                for pulses in (1, 4):
                    key = '%s_s%d_e%g_d%g_p%d' % (psf_type, steps, exc, dep, pulses)
                    r = st.psf_report(psf_type, exc, dep, steps, pulses,
                                      verbose=False)
                    n = r['psfs']['sted'].shape[1]
                    sc = [r['resolution_improvement_descanned'],
                          r.get('resolution_improvement_rescanned', np.nan),
                          r['excitation_dose'], r['depletion_dose'],
                          r['expected_emission'], r['pulses_per_position']]
                    out[key + '/scalars'] = np.array(sc, dtype=np.float64)
                    if psf_type == 'line':
                        sigma = steps / (2 * np.sqrt(2 * np.log(2)))
                        w, _ = st.get_width(r['psfs']['sted'][0, n // 2, :])
                        out[key + '/ratio'] = np.array(
                            [(sigma / w) ** 2 + 1,
                             int(np.round((sigma / w) ** 2 + 1))])
                    if pulses == 1:
                        for k, v in r['psfs'].items():
                            if steps <= 12:
                                out[key + '/psf/' + k] = v
                            else:     # central row + central column only
                                out[key + '/row/' + k] = v[0, n // 2, :]
                                out[key + '/col/' + k] = v[0, :, n // 2]
                    cases.append(key)
    out['cases'] = np.array(cases)
    save('g1_psf_report.npz', **out)


def g2_get_width():
    rows, widths, fits = [], [], []
    for psf_type, exc, dep, steps in (
            ('point', 0.25, 0, 4), ('point', 0.25, 9, 8), ('point', 4, 27, 12),
            ('line', 0.25, 0, 6), ('line', 1, 3, 12), ('line', 0.25, 9, 25),
            ('point', 0.01, 0, 25), ('line', 4, 27, 25)):
        r = st.psf_report(psf_type, exc, dep, steps, 1, verbose=False)
        n = r['psfs']['sted'].shape[1]
        for k in ('excitation', 'sted') + (('rescan_sted', 'descan_sted')
                                            if psf_type == 'line' else ()):
            row = r['psfs'][k][0, n // 2, :]
            w, f = st.get_width(row)
            rows.append(np.pad(row, (0, 129 - n)))
            fits.append(np.pad(f, (0, 129 - n)))
            widths.append([n, w])
    save('g2_get_width.npz', rows=np.array(rows), fits=np.array(fits),
         n_and_width=np.array(widths))


FIG2 = {  # inputs hand-tuned by the reference's author, line_sted_figure_2.py:77-162
    '1p0x_ld': (0.99, 0.99, 4, 4, 'descanned', 1, 0.01),
    '1p0x_lr': (0.99, 1.38282445, 4, 4, 'rescanned', 2, 0.01),
    '1p5x_ld': (1.5, 2.68125, 4, 2.825, 'descanned', 3, 0.25),
    '1p5x_lr': (1.5, 2.95425, 4, 2.618, 'rescanned', 3, 0.25),
    '2p0x_ld': (2., 4.04057, 4, 3.007, 'descanned', 4, 0.25),
    '2p0x_lr': (2., 4.07614, 4, 3.0227, 'rescanned', 4, 0.25),
    '2p5x_ld': (2.5, 5.13325, 4, 3.792, 'descanned', 6, 0.25),
    '2p5x_lr': (2.5, 5.15129, 4, 3.8, 'rescanned', 6, 0.25),
    '3p0x_ld': (3., 5.94563, 4, 5.034, 'descanned', 8, 0.25),
    '3p0x_lr': (3., 5.95587, 4, 5.0385, 'rescanned', 8, 0.25),
    '4p0x_ld': (4., 7.8386627, 4, 7.371, 'descanned', 10, 0.25),
    '4p0x_lr': (4., 7.840982, 4, 7.37195, 'rescanned', 10, 0.25),
}
TUNE_KEYS = ('excitation_brightness', 'depletion_brightness',
             'pulses_per_position', 'steps_per_excitation_psf_width',
             'excitation_dose', 'depletion_dose', 'expected_emission',
             'resolution_improvement_descanned')


def _tune_vec(r):
    v = [float(r[k]) for k in TUNE_KEYS]
    v.append(float(r.get('resolution_improvement_rescanned', np.nan)))
    return np.array(v)


def g3_tune_psf(which=None):
    """tune_psf at the fig-2 operating points (ref :365-476)."""
    out = {'keys': np.array(TUNE_KEYS + ('resolution_improvement_rescanned',))}
    for name, (pr, lr, pe, le, scan, nori, maxexc) in FIG2.items():
        if which and name not in which:
            continue
        t = time.time()
        point = st.tune_psf('point', 'descanned', float(pr), float(pe),
                            max_excitation_brightness=maxexc,
                            steps_per_improved_psf_width=4.)
        line = st.tune_psf('line', scan, float(lr), float(le),
                           max_excitation_brightness=maxexc,
                           steps_per_improved_psf_width=4.)
        out[name + '/inputs'] = np.array([pr, lr, pe, le, nori, maxexc,
                                          scan == 'rescanned'], dtype=np.float64)
        out[name + '/point'] = _tune_vec(point)
        out[name + '/line'] = _tune_vec(line)
        print(name, '%.1fs' % (time.time() - t), out[name + '/point'][:3],
              out[name + '/line'][:3], flush=True)
    save('g3_tune_psf.npz', **out)


def g4_conv():
    """H / H_t conventions (ref :567-594) on odd, even, 1-row and oversize PSFs."""
    rng = np.random.default_rng(20171003)
    out = {}
    cases = (('odd', (3, 40, 48), (1, 9, 11)), ('even', (3, 40, 48), (1, 8, 10)),
             ('row', (3, 40, 48), (1, 1, 7)), ('big', (2, 6, 5), (1, 9, 11)),
             ('two', (1, 33, 31), (1, 7, 7)))
    for name, xs, ps in cases:
        x = rng.random(xs)
        npsf = 3 if name == 'two' else 1
        psfs = [rng.random(ps) for _ in range(npsf)]
        d = st.Deconvolver(psfs, output_prefix='/tmp/_golden_tmp/', verbose=False)
        Hx = d.H(x)
        y = [rng.random(xs) for _ in range(npsf)]
        Ht_raw = d.H_t(y, normalize=False)
        Ht = d.H_t(y, normalize=True)
        out[name + '/x'] = x
        out[name + '/psfs'] = np.array(psfs)
        out[name + '/y'] = np.array(y)
        out[name + '/H'] = np.array(Hx)
        out[name + '/Ht_raw'] = Ht_raw
        out[name + '/Ht'] = Ht
        out[name + '/norm'] = d.H_t_normalization
    save('g4_conv.npz', **out)


def _objects():
    return {n: np_tif.tif_to_array(os.path.join(REF, 'test_object_%s.tif' % n))
            for n in ('cat', 'astronaut', 'lines', 'rings')}


def _fig2_psfs(name):
    """point_sted_psf / line_sted_psfs exactly as line_sted_figure_2.py:183-248."""
    import line_sted_figure_2 as f2
    pr, lr, pe, le, scan, nori, maxexc = FIG2[name]
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        c = f2.psf_comparison_pair(float(pr), float(lr), float(pe), float(le),
                                   scan, nori, max_excitation_brightness=maxexc)
    return c


def g8_fig2_psfs():
    out = {}
    for name in ('1p0x_ld', '1p5x_lr', '2p0x_lr'):
        c = _fig2_psfs(name)
        out[name + '/point_sted_psf'] = np.array(c['point_sted_psf'])
        out[name + '/line_sted_psfs'] = np.array(c['line_sted_psfs'])
        out[name + '/point'] = _tune_vec(c['point'])
        out[name + '/line'] = _tune_vec(c['line'])
    save('g8_fig2_psfs.npz', **out)


def g1b_line_dumps():
    """psf_report('line', ..., output_dir=...) (ref:311-346): the nine files the reference writes, and the two arrays
    among them that are not in the returned dict -- emission_psf.tif and sted_psf_line_rescan_unscaled.tif -- as read
    back with the reference's own np_tif (float32 on disk)."""
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix='_golden_dump_')
    args = ('line', 0.25, 9.0, 8, 1)
    st.psf_report(*args, verbose=False, output_dir=tmp)
    names = sorted(os.listdir(tmp))
    out = {'args': np.array(args[1:], dtype=np.float64), 'files': np.array(names)}
    for n in ('emission_psf.tif', 'sted_psf_line_rescan_unscaled.tif', 'sted_psf_line_rescan.tif'):
        out[n] = np_tif.tif_to_array(os.path.join(tmp, n))
    shutil.rmtree(tmp)
    save('g1b_line_dumps.npz', **out)


def g8b_fig2_psfs_more():
    """The two remaining line-rescan doses of BASELINE config 2 (2.5x: 6 orientations, 3.0x: 8), same layout as G8."""
    out = {}
    for name in ('2p5x_lr', '3p0x_lr'):
        c = _fig2_psfs(name)
        out[name + '/point_sted_psf'] = np.array(c['point_sted_psf'])
        out[name + '/line_sted_psfs'] = np.array(c['line_sted_psfs'])
        out[name + '/point'] = _tune_vec(c['point'])
        out[name + '/line'] = _tune_vec(c['line'])
    save('g8b_fig2_psfs_more.npz', **out)


def g5_rl():
    """simulate (+noise from numpy's legacy RNG, seed 0) and RL estimates."""
    objs = _objects()
    save('objects.npz', **objs)
    psfs = np.load(os.path.join(HERE, 'g8_fig2_psfs.npz'))
    out = {}
    runs = (('rings_point_1p5x', 'rings', psfs['1p5x_lr/point_sted_psf'], (1, 2, 5, 20, 100)),
            ('rings_line4_2p0x', 'rings', psfs['2p0x_lr/line_sted_psfs'], (1, 5, 20)),
            ('cat_line1_1p0x', 'cat', psfs['1p0x_ld/line_sted_psfs'], (5,)))
    for name, obj_name, psf_set, ks in runs:
        obj = objs[obj_name].astype(np.float64)
        d = st.Deconvolver(list(psf_set), output_prefix='/tmp/_golden_tmp/',
                           verbose=False)
        d.create_data_from_object(obj, total_brightness=5e10, random_seed=0)
        out[name + '/noiseless'] = np.array(d.noiseless_measurement)
        out[name + '/noisy'] = np.array(d.noisy_measurement)
        for k in range(1, max(ks) + 1):
            d.iterate()
            if k in ks:
                out[name + '/estimate_%d' % k] = d.estimate.copy()
        out[name + '/norm'] = d.H_t_normalization
        out[name + '/psf_key'] = np.array(
            {'rings_point_1p5x': '1p5x_lr/point_sted_psf',
             'rings_line4_2p0x': '2p0x_lr/line_sted_psfs',
             'cat_line1_1p0x': '1p0x_ld/line_sted_psfs'}[name])
        out[name + '/object'] = np.array(obj_name)
        print(name, 'done', flush=True)
    save('g5_rl.npz', **out)


def g9_progress():
    out = {}
    for n in (0, 1, 2, 3, 5, 17, 1025):
        out['n%d' % n] = np.array([flag for _, flag in
                                   st.logarithmic_progress(range(n), verbose=False)])
    save('g9_progress.npz', **out)


def g10_quality():
    """(a) record_iteration()'s history files as the reference writes them (float32 TIFF stacks,
    line_sted_tools.py:533-547), read back with the reference's np_tif; (b) the Fourier error
    map and its line profiles -- line_sted_figure_2.py:353-390 lives inside a figure function
    that cannot be called piecewise, so its numpy/scipy calls are issued here on the same data."""
    from scipy.ndimage import map_coordinates, gaussian_filter
    objs = _objects()
    psfs = np.load(os.path.join(HERE, 'g8_fig2_psfs.npz'))
    obj = objs['rings'].astype(np.float64)
    prefix = '/tmp/_golden_tmp/q_'
    d = st.Deconvolver(list(psfs['1p5x_lr/point_sted_psf']), output_prefix=prefix, verbose=False)
    d.create_data_from_object(obj, total_brightness=5e10, random_seed=0)
    for _, flag in st.logarithmic_progress(range(9), verbose=False):   # saves after iterations 2,3,5,9
        d.iterate()
        if flag:
            d.record_iteration()
    out = {'saved_iterations': np.array(d.saved_iterations),
           'noisy': np.array(d.noisy_measurement),
           'true_object': d.true_object,
           'estimate': d.estimate.copy(),
           'estimate_history_tif': np_tif.tif_to_array(prefix + 'estimate_history.tif'),
           'ft_error_history_tif': np_tif.tif_to_array(prefix + 'estimate_FT_error_history.tif')}
    est, true_object = d.estimate[0], d.true_object[0]
    fe = np.abs(np.fft.fftshift(np.fft.fftn(est - true_object))) / np.prod(true_object.shape)   # :353-355
    out['fourier_error'] = fe
    n_x, n_y = true_object.shape                                                                 # :363
    samps, rad = 1000, 0.3
    for tag, ang_deg in (('best', 0.0), ('worst', 90 / 4)):                                      # :364,370
        ang = ang_deg * 2 * np.pi / 360
        x0, x1 = (0.5 + rad * np.array((-np.cos(ang), np.cos(ang)))) * n_x
        y0, y1 = (0.5 + rad * np.array((-np.sin(ang), np.sin(ang)))) * n_y
        xy = np.vstack((np.linspace(x0, x1, samps), np.linspace(y0, y1, samps)))                 # :379
        z = map_coordinates(np.transpose(fe), xy)                                                # :383
        out['profile_raw_' + tag] = z
        out['profile_' + tag] = gaussian_filter(z, sigma=samps / 80)                             # :387
        out['angle_' + tag] = np.array(ang_deg)
    save('g10_quality.npz', **out)


if __name__ == '__main__':
    todo = sys.argv[1:] or ['g1', 'g2', 'g4', 'g9', 'g8', 'g5', 'g3']
    os.makedirs('/tmp/_golden_tmp', exist_ok=True)
    for t in todo:
        {'g1': g1_psf_report, 'g2': g2_get_width, 'g3': g3_tune_psf,
         'g4': g4_conv, 'g5': g5_rl, 'g8': g8_fig2_psfs, 'g8b': g8b_fig2_psfs_more, 'g1b': g1b_line_dumps,
         'g9': g9_progress, 'g10': g10_quality}[t]()
