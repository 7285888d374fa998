"""Manual helper (not a test): BASELINE config 5's tolerance study -- how far the Richardson-Lucy estimate of
(i) the float32 plan, (ii) a float32 plan whose spectra are rounded to IEEE half precision on their way to
memory (per-spectrum power-of-two scale) and (iii) one rounded to bfloat16 drift from the float64 plan's,
iteration by iteration (default: the 4096 x 4096 tile, synthetic uniform object, point-descan PSF, K = 100).

(ii) and (iii) are study builds of the library (`python -m rescan_line_sted_amd._build --variant q16|qbf16`,
conv_kernels.hpp RL_SPEC_QUANT): they measure the numerical effect of 16-bit spectrum storage between the row
and the column kernels, not its bandwidth.  Each build runs in a process of its own (RLSTED_LIB); the float64
checkpoints travel through /tmp.

    python tools/gpu/gpu_tolerance_study.py [n] [K] [out.json]      (default gpurun_out/r03/tolerance_study_<n>.json)
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
MARKS = [1, 2, 3, 5, 8, 10, 15, 20, 30, 40, 50, 60, 80, 100]


def setup(n):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
    psf = [g['2p0x_lr/point_sted_psf'][0]]
    obj = np.random.default_rng(4321).random((1, n, n)) * 255
    return psf, obj, 5e10 * (n / 128) ** 2


def worker(mode, n, K, tmp):
    from rescan_line_sted_amd import _lib
    psf, obj, brightness = setup(n)
    marks = [m for m in MARKS if m <= K]
    if mode == 'f64':
        plan = _lib.DeconvPlan(psf, 1, n, n, dtype='f64')
        plan.set_object(obj, brightness)
        plan.simulate(seed=5)
        np.save(os.path.join(tmp, 'noisy.npy'), plan.measurement())
        done = 0
        for k in marks:
            plan.iterate(k - done)
            done = k
            np.save(os.path.join(tmp, 'ref_%d.npy' % k), plan.estimate()[0])
        return
    plan = _lib.DeconvPlan(psf, 1, n, n, dtype='f32')
    plan.set_object(obj, brightness)
    plan.set_measurement(np.load(os.path.join(tmp, 'noisy.npy')))
    rows, done = [], 0
    for k in marks:
        plan.iterate(k - done)
        done = k
        ref, est = np.load(os.path.join(tmp, 'ref_%d.npy' % k)), plan.estimate()[0]
        d = np.abs(est - ref)
        rows.append({'iteration': k, 'max_over_max': float(d.max() / ref.max()),
                     'rms_over_rms': float(np.sqrt((d ** 2).mean()) / np.sqrt((ref ** 2).mean()))})
        print(mode, rows[-1], flush=True)
    json.dump(rows, open(os.path.join(tmp, 'rows_%s.json' % mode), 'w'))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--worker':
        return worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    tmp = os.path.join('/tmp', 'rlsted_tolerance_%d' % os.getpid())
    os.makedirs(tmp, exist_ok=True)
    psf, obj, brightness = setup(n)
    psf_sum = max(1.0, float(psf[0].sum()))
    # DC term of an estimate-type spectrum <= total brightness x sum(psf); of a ratio-type spectrum ~ pixels x sum(psf)
    q_est = int(np.ceil(np.log2(brightness * psf_sum))) + 1
    q_ratio = int(np.ceil(np.log2(n * n * psf_sum))) + 2
    libdir = os.path.join(ROOT, 'rescan_line_sted_amd', '_lib')
    out = {'shape': [n, n], 'psf': '2p0x_lr/point_sted_psf', 'object': 'default_rng(4321).random * 255', 'K': K,
           'fp16_scale_exponents': {'estimate_spectra': q_est, 'ratio_spectra': q_ratio},
           'what': 'max|a-b|/max|b| and rms ratio of the f32 plan (and of f32 plans with spectra rounded to fp16 / bf16 on '
                   'their way to memory) against the f64 plan, same noisy measurement'}
    for mode, lib in (('f64', None), ('f32', None), ('fp16_spectra', 'librlsted_q16.so'), ('bf16_spectra', 'librlsted_qbf16.so')):
        env = dict(os.environ, RLSTED_Q_EXP_EST=str(q_est), RLSTED_Q_EXP_RATIO=str(q_ratio))
        if lib:
            env['RLSTED_LIB'] = os.path.join(libdir, lib)
            if not os.path.exists(env['RLSTED_LIB']):
                out[mode] = 'library variant not built'
                continue
        subprocess.check_call([sys.executable, os.path.abspath(__file__), '--worker', mode, str(n), str(K), tmp], env=env)
        if mode != 'f64':
            out[mode] = json.load(open(os.path.join(tmp, 'rows_%s.json' % mode)))
    dest = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, 'gpurun_out', 'r04', 'tolerance_study_%d.json' % n)
    os.makedirs(os.path.dirname(os.path.abspath(dest)), exist_ok=True)
    json.dump(out, open(dest, 'w'), indent=1)
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)


if __name__ == '__main__':
    main()
