"""CPU-side checks of the product: the C-ABI library loads and exports every
symbol include/rlsted.h declares (no compute calls without a GPU), error
conventions, and the host-side logic (Gaussian fit, Brent search, progress
iterator, drop-in shims)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, max_rel


@pytest.fixture(scope='module')
def lib():
    from rescan_line_sted_amd import _lib
    return _lib


def _declared_functions():
    text = open(os.path.join(ROOT, 'include', 'rlsted.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rl_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol(lib):
    declared = _declared_functions()
    assert len(declared) >= 25
    raw = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), 'librlsted.so lacks ' + name
    assert sorted(lib.PROTOTYPES) == declared          # the ctypes table binds exactly the header


def test_library_asks_for_eight_hardware_queues_unless_told_otherwise():
    """Loading librlsted.so sets GPU_MAX_HW_QUEUES=8 in the process environment (the HIP runtime reads it at its first call: the
    sweep's contexts then overlap on 8 hardware queues instead of 4, csrc/rlsted.cpp rl_runtime_defaults) and leaves a value the
    user chose alone.  In child processes: the environment of this one is already decided."""
    import subprocess
    import sys
    code = ("import ctypes, os, sys; sys.path.insert(0, %r); from rescan_line_sted_amd import _lib; "
            "g = ctypes.CDLL(None).getenv; g.restype = ctypes.c_char_p; print(g(b'GPU_MAX_HW_QUEUES').decode())" % ROOT)
    env = {k: v for k, v in os.environ.items() if k != 'GPU_MAX_HW_QUEUES'}
    assert subprocess.check_output([sys.executable, '-c', code], env=env, text=True).strip() == '8'
    env['GPU_MAX_HW_QUEUES'] = '2'
    assert subprocess.check_output([sys.executable, '-c', code], env=env, text=True).strip() == '2'


def test_no_gpu_errors_are_loud(lib):
    if lib.device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(lib.RlstedError) as e:
        lib.Context(0)
    assert 'librlsted error -2' in str(e.value)         # RL_ERR_HIP + message, no silent fallback
    assert lib.lib.rl_version() >= 100
    assert [lib.lib.rl_fft_length_for(n) for n in (1, 64, 65, 181, 213, 565, 1100, 2101, 2305, 4609)] == \
           [64, 64, 192, 192, 256, 576, 1152, 2304, 4608, 0]


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'rescan_line_sted_amd')
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.cpp', '.hip', '.hpp', '.h')):
                text = open(os.path.join(base, f), errors='replace').read()
                assert not re.search(r'^\s*(from|import)\s+oracle', text, flags=re.M), f
                assert '/root/reference' not in text, f
    for f in ('bench.py', '__graft_entry__.py'):
        assert '/root/reference' not in open(os.path.join(ROOT, f)).read()


def test_gauss_fit_matches_reference_widths(golden):
    from rescan_line_sted_amd import psf
    g = golden('g2_get_width')
    for row, fit, (n, w) in zip(g['rows'], g['fits'], g['n_and_width']):
        n = int(n)
        s, f = psf.get_width(row[:n])
        assert abs(s) == pytest.approx(abs(w), rel=1e-8)
        assert max_rel(f, fit[:n]) < 1e-7 and f.shape == (n,)
    with pytest.raises(Exception):
        psf.get_width(np.zeros(2))


def test_scalar_minimizer_follows_scipy_default_method():
    from scipy.optimize import minimize_scalar
    from rescan_line_sted_amd.psf import _ScalarMinimizer
    funcs = [lambda x: (x - 2.0) ** 2, lambda x: (abs(x) - 1.5) ** 2 + 0.1 * x, lambda x: np.cosh(x - 0.3),
             lambda x: (np.sqrt(abs(x)) - 1.3) ** 2, lambda x: (x + 7.0) ** 4 + x]
    for f in funcs:
        m = _ScalarMinimizer(f)
        x = m.minimize()
        ref = minimize_scalar(f)
        assert x == ref.x and m.calls == ref.nfev          # same iterates, same evaluation count


def test_logarithmic_progress(golden, capsys):
    from rescan_line_sted_amd import line_sted_tools as st
    g = golden('g9_progress')
    for n in (0, 1, 2, 3, 5, 17, 1025):
        got = list(st.logarithmic_progress(range(n), verbose=False))
        assert [x for x, _ in got] == list(range(n))
        assert [f for _, f in got] == list(g['n%d' % n])
    assert list(st.logarithmic_progress([])) == []          # a generator that yields nothing, as in the reference


def test_dropin_shims_resolve_the_reference_module_names():
    code = ("import sys; sys.path.insert(0, %r); import line_sted_tools as st, np_tif;"
            "print(all(hasattr(st, n) for n in ('psf_report','generate_psfs','tune_psf','Deconvolver',"
            "'logarithmic_progress','get_width')), hasattr(np_tif,'tif_to_array'), hasattr(np_tif,'array_to_tif'))"
            % os.path.join(ROOT, 'dropin'))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd='/tmp')
    assert out.stdout.split() == ['True', 'True', 'True'], out.stderr


def test_deconvolver_surface_matches_reference(tmp_path):
    from rescan_line_sted_amd import line_sted_tools as st
    import inspect
    sig = inspect.signature(st.Deconvolver.__init__)
    assert list(sig.parameters)[:4] == ['self', 'psfs', 'output_prefix', 'verbose']
    assert list(inspect.signature(st.Deconvolver.create_data_from_object).parameters) == \
        ['self', 'obj', 'total_brightness', 'random_seed']
    for name in ('load_data_from_tif', 'iterate', 'record_iteration', 'record_data', 'H', 'H_t'):
        assert callable(getattr(st.Deconvolver, name))
    assert list(inspect.signature(st.Deconvolver.H_t).parameters) == ['self', 'y', 'normalize']
    d = st.Deconvolver([np.ones((1, 3, 3))], output_prefix=str(tmp_path / 'sub') + '/', verbose=False)
    assert os.path.isdir(str(tmp_path / 'sub'))             # ref:487-488 creates dirname(prefix)
    assert (d.num_iterations, d.saved_iterations, d.estimate_history) == (0, [], [])
    with pytest.raises(AttributeError):
        d.estimate
    from rescan_line_sted_amd import psf
    for fn, names in ((psf.psf_report, ['psf_type', 'excitation_brightness', 'depletion_brightness',
                                        'steps_per_excitation_psf_width', 'pulses_per_position', 'verbose', 'output_dir']),
                      (psf.generate_psfs, ['shape', 'excitation_brightness', 'depletion_brightness', 'blur_sigma',
                                           'psf_type', 'output_dir', 'verbose']),
                      (psf.tune_psf, ['psf_type', 'scan_type', 'desired_resolution_improvement',
                                      'desired_emissions_per_molecule', 'max_excitation_brightness',
                                      'steps_per_improved_psf_width', 'relative_error', 'verbose_results',
                                      'verbose_iterations'])):
        assert list(inspect.signature(fn).parameters) == names


def test_pmc_traffic_json_is_what_the_tool_makes_of_the_committed_counter_files(tmp_path):
    """profiles/r04/pmc_traffic.json (read by bench.py for roofline.traffic) is reproducible from the committed rocprofv3
    counter files with tools/pmc_traffic.py; its per-iteration sum is what DESIGN.md quotes (12.9 MB per frame-iteration,
    1.76x the algorithmic bytes)."""
    import json
    import subprocess
    import sys
    prof = os.path.join(ROOT, 'profiles', 'r04')
    committed = json.load(open(os.path.join(prof, 'pmc_traffic.json')))
    out = tmp_path / 'traffic.json'
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'pmc_traffic.py'),
                           os.path.join(prof, 'pmc_fetch_size.csv'), os.path.join(prof, 'pmc_write_size.csv'),
                           str(out), str(committed['frames_per_launch']), '512', '1'], stdout=subprocess.DEVNULL)
    again = json.load(open(out))
    assert again == committed
    it = committed['rl_iteration']
    assert it['algorithmic_bytes'] == 4 * 512 * 512 * 7 * committed['frames_per_launch']
    assert 1.5 < it['ratio'] < 2.0
    # the doubled FETCH_SIZE reproduces the bytes rowpass_FWD must read (32 frames x 512 x 512 x 4 bytes) to within 2 %
    fwd = committed['rowpass_FWD']
    assert abs(2 * fwd['fetch_kb_reported'] * 1024 / (committed['frames_per_launch'] * 512 * 512 * 4) - 1) < 0.02


def test_psfs_of_different_shapes_are_embedded_on_a_common_centre():
    """One plan holds one (V, py, px) stack; the reference convolves with each PSF on its own (ref:573,585).
    The centred zero embedding leaves the 'same'-mode convolution unchanged (odd, even, 1-wide shapes)."""
    from rescan_line_sted_amd import _lib
    from oracle import line_sted_oracle as orc
    rng = np.random.default_rng(0)
    psfs = [rng.random((1, 4, 9)), rng.random((1, 7, 2)), rng.random((1, 1, 6))]
    x = rng.random((2, 20, 23))
    common = _lib.common_psf_shape(psfs)
    assert len({q.shape for q in common}) == 1 and common[0].shape[1] % 2 == 1 and common[0].shape[2] % 2 == 1
    for p, q in zip(psfs, common):
        assert np.abs(orc.conv_same(x, p) - orc.conv_same(x, q)).max() < 1e-13
    same = [rng.random((1, 5, 5)) for _ in range(2)]
    assert _lib.common_psf_shape(same) is same


def test_default_path_kernels_do_not_spill(tmp_path):
    """DESIGN.md's statements about scratch are checked against the ISA of HEAD.  fft_kernels.hip is
    compiled device-only for L = 576, 2304 and 4608 (the flags of _build.py) and `.private_segment_fixed_size` is read from
    the kernel descriptors: 0 for every f32 kernel of the 512 x 512 path, for the split column pass and for the f32 row kernels
    of the long lengths; the two whole-pass outer column kernels keep the small, documented amounts DESIGN.md section 3 states
    (L = 2304 on 16-column tiles: <= 96 bytes per lane; L = 4608: <= 24)."""
    import re
    import shutil
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    from rescan_line_sted_amd import _build
    if not (shutil.which(_build.HIPCC) or os.path.exists(_build.HIPCC)):
        pytest.skip('no hipcc')
    src = os.path.join(_build.CSRC, 'fft_kernels.hip')

    def scratch(L):
        out = str(tmp_path / ('fft_%d.s' % L))
        subprocess.check_call([_build.HIPCC] + _build.COMMON + _build.DEVICE + _build.FFT_FLAGS +
                              ['-DRL_CFG_L=%d' % L, '--cuda-device-only', '-S', src, '-o', out], stderr=subprocess.DEVNULL)
        txt = open(out).read()
        names = subprocess.run(['c++filt'], input='\n'.join(re.findall(r'\.name:\s+(\S+)', txt)), capture_output=True, text=True).stdout.split('\n')
        priv = [int(x) for x in re.findall(r'\.private_segment_fixed_size:\s+(\d+)', txt)]
        return dict(zip(names, priv))
    with ThreadPoolExecutor(3) as ex:
        s576, s2304, s4608 = ex.map(scratch, (576, 2304, 4608))
    f32 = {k: v for k, v in s576.items() if 'float' in k and 'double' not in k}
    assert len(f32) > 20 and all(v == 0 for v in f32.values()), {k: v for k, v in f32.items() if v}
    assert all(v == 0 for k, v in s576.items() if 'k_rowpair' in k or 'k_colconv<576, 4, 0' in k)      # the f64 RL loop of the headline size too
    for table, L, whole_max in ((s2304, 2304, 96), (s4608, 4608, 24)):
        for k, v in table.items():
            if 'double' in k:
                if 'k_colconv_outer' in k:
                    assert v == 0, (k, v)     # float64 column pass on the outer-decimation body (round 4): 218 registers at L = 2304, ~415 with accumulation registers at L = 4608, nothing in scratch
                continue                      # (the other f64 kernels of the long lengths: DESIGN.md section 8, follow-ups)
            if 'k_colconv_outer' in k and re.search(r', (true|false), 0, float>', k):
                assert v <= whole_max, (k, v)  # the whole pass (single-view plans)
            elif 'k_colconv_outer' in k and ', true, ' in k:
                assert v == 0, (k, v)          # the split pass with the real multiplier (what the reference's PSFs run)
            elif 'k_rowpair' in k or ('k_rowpass' in k and 'true, float, true' not in k):
                assert v == 0, (k, v)
