"""Manual helper (not a test): BASELINE config 2 end to end on one GPU -- the astronaut at 512x512
(np.kron x4), point-descan vs line-rescan at the four STED doses 1p5x / 2p0x / 2p5x / 3p0x
(line_sted_figure_2.py:93-148; 3, 4, 6, 8 line orientations), 20 RL iterations.  The PSF sets are
computed here by the product (tune_psf + psf_report + rotation on the device), then every
(dose, mode) is run as a batch of frames; one frame per case is checked against the CPU oracle
given the device's own noisy measurement.  Writes gpurun_out/config2.json."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_line_sted_amd import _lib, psf
from oracle import line_sted_oracle as orc
objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
obj = np.kron(objs['astronaut'].astype(np.float64), np.ones((1, 4, 4)))[0]
K, B, TB = 20, 64, 5e10 * 16
check = '--no-check' not in sys.argv
rows = []
t0 = time.perf_counter()
sets, comparisons = psf.figure_2_psfs([d + '_lr' for d in ('1p5x', '2p0x', '2p5x', '3p0x')])
t_psf = time.perf_counter() - t0
print('PSF sets for 4 doses (tune_psf x8, psf_report, rotations) on the device: %.2f s' % t_psf, flush=True)
for name, psfs in sets.items():
    psfs = [np.asarray(p) for p in psfs]
    V = len(psfs)
    plan = _lib.DeconvPlan(psfs, B, 512, 512, dtype='f32')
    plan.set_object(np.broadcast_to(obj, (B, 512, 512)), TB)
    plan.bench_cycles(K, 1, seed=1)
    t0 = time.perf_counter(); plan.bench_cycles(K, 3, seed=2); el = time.perf_counter() - t0
    fps = 3 * B / el
    alg = 4 * 512 * 512 * ((2 * V + 2) + K * (3 * V + 4))
    row = {'case': name, 'views': V, 'frames_per_s': fps, 'algorithmic_MB_per_frame': alg / 1e6, 'roofline_frac': alg * fps / 8e12}
    if check:
        noisy = plan.measurement()[0]                       # (V, 512, 512), the device's Philox draw
        d = orc.Deconvolver(psfs)
        d.true_object = obj[None]
        d.noisy_measurement = [noisy[v][None] for v in range(V)]
        for _ in range(K):
            d.iterate()
        est = plan.estimate()[0]
        row['max_rel_err_vs_oracle'] = float(np.abs(est - d.estimate[0]).max() / d.estimate[0].max())
    rows.append(row)
    print(json.dumps(row), flush=True)
    del plan
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump({'psf_seconds': t_psf, 'frames_per_plan': B, 'rl_iters': K, 'rows': rows}, open(os.path.join(ROOT, 'gpurun_out', 'config2.json'), 'w'), indent=1)
