"""Build librlsted.so (hipcc, gfx950) in-tree.

    python -m rescan_line_sted_amd._build [--force]

Objects go to build/ (git-ignored), the library to rescan_line_sted_amd/_lib/.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(ROOT, 'build', 'obj')
LIBDIR = os.path.join(HERE, '_lib')
LIB = os.path.join(LIBDIR, 'librlsted.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'
FFT_LENGTHS = (64, 192, 256, 576, 1152, 2304, 4608)
COMMON = ['-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include')]
DEVICE = ['--offload-arch=' + ARCH, '-munsafe-fp-atomics'] + os.environ.get('RL_EXTRA_HIPCC', '').split()
# The butterflies are written on (re, im) pairs; LLVM's SLP vectoriser re-pairs them into
# v_pk_* ops at the price of ~200 v_mov shuffles per transform: measured 5-6 % slower.
FFT_FLAGS = ['-fno-slp-vectorize']


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hpp', '.h'))] + \
           [os.path.join(ROOT, 'include', 'rlsted.h')]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError('build failed: %s\n%s' % (' '.join(cmd), r.stdout))
    return r.stdout


def jobs():
    out = []
    for L in FFT_LENGTHS:
        out.append((os.path.join(OBJ, 'fft_%d.o' % L), os.path.join(CSRC, 'fft_kernels.hip'),
                    DEVICE + FFT_FLAGS + ['-DRL_CFG_L=%d' % L]))
    out.append((os.path.join(OBJ, 'aux_kernels.o'), os.path.join(CSRC, 'aux_kernels.hip'), DEVICE))
    out.append((os.path.join(OBJ, 'psf_kernels.o'), os.path.join(CSRC, 'psf_kernels.hip'), DEVICE))
    out.append((os.path.join(OBJ, 'rlsted.o'), os.path.join(CSRC, 'rlsted.cpp'), ['-x', 'hip'] + DEVICE))
    out.append((os.path.join(OBJ, 'comm.o'), os.path.join(CSRC, 'comm.cpp'), ['-x', 'hip'] + DEVICE))
    out.append((os.path.join(OBJ, 'psf_api.o'), os.path.join(CSRC, 'psf_api.cpp'), ['-x', 'hip'] + DEVICE))
    out.append((os.path.join(OBJ, 'quality_kernels.o'), os.path.join(CSRC, 'quality_kernels.hip'), DEVICE))
    out.append((os.path.join(OBJ, 'sep_kernels.o'), os.path.join(CSRC, 'sep_kernels.hip'), DEVICE))
    out.append((os.path.join(OBJ, 'fig3_kernels.o'), os.path.join(CSRC, 'fig3_kernels.hip'), DEVICE))
    out.append((os.path.join(OBJ, 'quality_api.o'), os.path.join(CSRC, 'quality_api.cpp'), ['-x', 'hip'] + DEVICE))
    out.append((os.path.join(OBJ, 'gauss_fit.o'), os.path.join(CSRC, 'gauss_fit.cpp'), ['-ffp-contract=off']))
    return [j for j in out if os.path.exists(j[1])]


VARIANTS = {   # study builds: librlsted_<name>.so beside the product library (python -m ..._build --variant NAME)
    'q16': ['-DRL_SPEC_QUANT=1'],     # spectra rounded to IEEE half on their way to memory (BASELINE config 5 study)
    'qbf16': ['-DRL_SPEC_QUANT=2'],   # ... to bfloat16
}


def build_variant(name, verbose=False, flags=None):
    """A study build of the whole library with extra defines, objects under build/obj_<name>/.  `flags`: an explicit flag
    list for a development A/B build (tools/ab_build.py); the named study variants above take theirs from VARIANTS."""
    global OBJ, LIB, DEVICE
    keep = (OBJ, LIB, DEVICE)
    try:
        OBJ = os.path.join(ROOT, 'build', 'obj_' + name)
        LIB = os.path.join(LIBDIR, 'librlsted_%s.so' % name)
        DEVICE = DEVICE + (VARIANTS[name] if flags is None else list(flags))
        return build(force=flags is not None, verbose=verbose)   # (a development build's flags change between runs)
    finally:
        OBJ, LIB, DEVICE = keep


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = _headers()
    todo = [(o, s, f) for o, s, f in jobs() if force or _stale(o, [s] + hdrs)]

    def compile_one(job):
        o, s, f = job
        _run([HIPCC] + COMMON + f + ['-c', s, '-o', o])
        return o
    if todo:
        with ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            for o in ex.map(compile_one, todo):
                if verbose:
                    print('compiled', os.path.relpath(o, ROOT))
    objs = [o for o, _, _ in jobs()]
    if force or todo or _stale(LIB, objs):
        _run([HIPCC, '-shared', '-fPIC', '--offload-arch=' + ARCH] + objs + ['-ldl', '-o', LIB])
        if verbose:
            print('linked', os.path.relpath(LIB, ROOT))
    return LIB


if __name__ == '__main__':
    if '--variant' in sys.argv:
        print(build_variant(sys.argv[sys.argv.index('--variant') + 1], verbose=True))
    else:
        print(build(force='--force' in sys.argv, verbose=True))
