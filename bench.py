#!/usr/bin/env python3
"""Headline benchmark: simulated frames/s, 512x512, 20 Richardson-Lucy iterations.

One "step" = one pass of the hot path over one batch of synthetic frames:
for every frame  noiseless = H(object); noisy = Poisson(noiseless) + 1e-9;
estimate = 1; 20 x { estimate *= H_t(noisy / H(estimate)) }   -- i.e.
Deconvolver.create_data_from_object + 20 x Deconvolver.iterate of the reference
(figure_generation/line_sted_tools.py:496-531).  Objects and PSF spectra are
resident in HBM when the timed region starts; nothing crosses PCIe inside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--dtype f32|f64] [--size 512|2048]

--gpus N > 1 without a launcher: this process starts `python -m torch.distributed.run
--nproc-per-node N` on itself (before anything touches the GPU) and relays the ranks' one JSON
line.  Under a launcher (RANK set) every rank runs B frames (weak scaling), there is no collective
in the data path, the ranks meet in a barrier before and after the K timed steps, the maximum time
over ranks counts, and ONE gather of the final estimates over RCCL / xGMI follows, timed separately.
The barrier, the reduction and the gather are the C ABI's (rl_comm_*, rl_gather*: RCCL loaded by
librlsted.so itself); torch.distributed is only the fallback transport should that fail to start.

--size 2048 is BASELINE config 3 (synthetic 2048x2048 object, line-rescan, 4 views); the default single-GPU run
also carries three steps of it as "size_2048" in the same JSON line (--no-2048 skips them).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

K_ITERS = 20
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ERROR_DEFINITION = ('normwise: max|a-b| / max|b| over the frame; pixelwise: max over pixels with b > 1e-3 max(b) of '
                    '|a-b| / b; a = device estimate, b = float64 oracle on the device-drawn noisy measurement')


def workload(size):
    """512: BASELINE config[1] restricted to the metric's quoted case -- astronaut 128x128 -> 512x512
    (np.kron x4), point-descan STED PSF of the 2.0x operating point (107x107,
    line_sted_figure_2.py:107-120,235-238), brightness 5e10*16.
    2048: config[2] -- default_rng(1234) uniform [0,255) object, the 4 line-rescan views of the same
    operating point, brightness 5e10*256 (the same ~1e7 counts per pixel)."""
    psfs = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
    if size == 512:
        objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
        obj = np.kron(objs['astronaut'].astype(np.float64), np.ones((1, 4, 4)))[0]
        psf = [psfs['2p0x_lr/point_sted_psf'][0]]
        name = ('astronaut 128->512x512 (np.kron x4), point-descan STED PSF 107x107 (2.0x operating point), '
                'simulate (H + Philox Poisson) + 20 RL iterations per frame')
        return obj, psf, 5e10 * 16, name
    obj = np.random.default_rng(1234).random((size, size)) * 255.0
    psf = [p[None] for p in psfs['2p0x_lr/line_sted_psfs'][:, 0]]
    name = ('default_rng(1234) uniform %dx%d object, line-rescan STED, 4 views 107x107 (2.0x operating point), '
            'simulate (H + Philox Poisson) + 20 RL iterations per frame (BASELINE config 3 shape)' % (size, size))
    return obj, psf, 5e10 * (size // 128) ** 2, name


def algorithmic_bytes_per_frame(n_pix, n_psf, k):
    # fp32 storage, each array touched once per logical pass
    return 4 * n_pix * ((2 * n_psf + 2) + k * (3 * n_psf + 4))


# ------------------------------------------------------------------ CPU baseline (oracle, float64)
def _cpu_worker(args):
    """One worker of the all-core leg: whole cycles of the oracle until the time budget is spent."""
    size, budget_s, seed0 = args
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle import line_sted_oracle as orc
    obj, psf, brightness, _ = workload(size)
    t0 = time.perf_counter()
    frames = 0
    while True:
        d = orc.Deconvolver(psf)
        d.create_data_from_object(obj[None].copy(), brightness, random_seed=seed0 + frames)
        for _ in range(K_ITERS):
            d.iterate()
        frames += 1
        if time.perf_counter() - t0 > budget_s:
            break
    return frames, time.perf_counter() - t0


def _usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:      # cgroup v2 CPU quota of the container, if any
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(size, budget_s=10.0):
    """The oracle (numpy float64 restatement of the reference) on a bounded sample of the same
    workload: (i) one core, as the reference's single-threaded scipy; (ii) one worker process per
    usable host core over independent frames.  Runs BEFORE this process
    touches the GPU (the workers are fresh `spawn` children)."""
    frames, el = _cpu_worker((size, budget_s, 0))
    out = {'value': frames / el, 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
           'sample': '%d frames of the same %dx%d / K=%d workload, numpy float64 oracle, 1 thread, %.1f s'
                     % (frames, size, size, K_ITERS, el)}
    cores = _usable_cores()
    if cores > 1:
        import multiprocessing as mp
        t0 = time.perf_counter()
        with mp.get_context('spawn').Pool(cores) as pool:
            res = pool.map(_cpu_worker, [(size, budget_s, 1000 * (i + 1)) for i in range(cores)])
        wall = time.perf_counter() - t0
        total = sum(f for f, _ in res)
        out['all_cores'] = {'value': total / max(t for _, t in res), 'unit': 'frames/s', 'cores': cores,
                            'os_cpu_count': os.cpu_count(), 'cpu_model': _cpu_model(),
                            'sample': '%d frames, one oracle process per core (%d), %.1f s each, %.1f s wall with start-up'
                                      % (total, cores, budget_s, wall)}
    return out


# ------------------------------------------------------------------ N > 1 plumbing
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv):
    """`python bench.py --gpus N` on its own: start N ranks with torch.distributed.run as a CHILD
    process (nothing in this process has touched the GPU; it never will) and relay their line."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + argv
    r = subprocess.run(cmd, stdout=subprocess.PIPE)
    lines = [ln for ln in r.stdout.decode(errors='replace').splitlines() if ln.startswith('{')]
    if lines:
        print(lines[-1], flush=True)
    return r.returncode if r.returncode else (0 if lines else 1)


class TorchComm:
    """Fallback transport (torch.distributed, backend nccl = RCCL) with sharding.RcclComm's interface."""

    def __init__(self, local_rank, backend='nccl'):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.cuda = backend == 'nccl'
        if self.cuda:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.dev = 'cuda' if self.cuda else 'cpu'

    def barrier(self):
        t = self.torch.zeros(1, device=self.dev)
        self.dist.all_reduce(t)
        if self.cuda:
            self.torch.cuda.synchronize()

    def allreduce_max(self, x):
        t = self.torch.tensor([float(x)], device=self.dev, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_plan(self, plan, counts, which='estimate', root=0, to_host=True):
        t = self.torch.as_tensor(plan.device_array(which), device=self.dev) if self.cuda else self.torch.from_numpy(plan.estimate())
        bufs = [self.torch.empty_like(t) for _ in range(self.world)] if self.rank == root else None
        self.dist.gather(t.contiguous(), bufs, dst=root)
        if self.cuda:
            self.torch.cuda.synchronize()
        return (0, sum(counts) * t[0].numel(), 0) if self.rank == root else (None, 0, 0)

    def close(self):
        self.dist.destroy_process_group()


class StubPlan:
    """RLSTED_BENCH_STUB=1 (tests only): stands in for the device plan so that the N > 1 control flow
    of this file -- self launch, rank environment, barriers, max over ranks, gather, the JSON line --
    runs on a box without GPUs.  Nothing it produces is a measurement."""

    def __init__(self, psf, B, ny, nx):
        self.B, self.V, self.ny, self.nx = B, len(psf), ny, nx
        self._est = np.ones((B, ny, nx))

    def set_object(self, *a):
        pass

    def bench_cycles(self, k, reps, seed=0):
        time.sleep(0.002 * reps)
        return 2.0 * reps

    def estimate(self):
        return self._est

    def measurement(self):
        return np.ones((self.B, self.V, self.ny, self.nx))

    def info(self):
        return {'ly': 0, 'lx': 0, 'pitch': self.nx // 2 + 8, 'device_bytes': 0}

    def time_cycle(self, k, seed=0):
        return {'colconv_H': (0.03, 168), 'rowpass_RATIO': (0.03, 160), 'colconv_Ht': (0.03, 160),
                'rowpass_UPDATE': (0.03, 160), 'rowpass_FWD': (0.02, 16), 'poisson': (0.1, 8)}, min(32, self.B)

    class ctx:
        @staticmethod
        def synchronize():
            pass


def make_comm(rank, world, local_rank, stub):
    if stub:
        return TorchComm(local_rank, backend='gloo'), 'torch.distributed gloo (stub)'
    try:
        from rescan_line_sted_amd import sharding
        return sharding.RcclComm(rank, world, device=local_rank), 'rl_comm (RCCL through the C ABI)'
    except Exception as exc:      # the fallback is reported in the output line
        return TorchComm(local_rank), 'torch.distributed nccl (rl_comm failed: %r)' % (exc,)


def accuracy(plan, psf, size, dtype):
    """One frame of the batch against the float64 oracle run on the SAME (device-drawn) noisy measurement."""
    from oracle import line_sted_oracle as orc
    meas = plan.measurement()[0]
    d = orc.Deconvolver(psf)
    d.noisy_measurement = [m[None].copy() for m in meas]
    for _ in range(K_ITERS):
        d.iterate()
    a, b = plan.estimate()[0], d.estimate[0]
    big = b > 1e-3 * b.max()
    return {'definition': ERROR_DEFINITION, 'dtype': dtype, 'frames_checked': 1, 'rl_iters': K_ITERS,
            'normwise': float(np.abs(a - b).max() / b.max()), 'pixelwise': float((np.abs(a - b)[big] / b[big]).max()),
            'contract': 1e-5 if dtype == 'f32' else 1e-10}


def second_size(size, B, steps, warmup, device):
    """`steps` whole cycles (simulate + 20 RL iterations) of `B` frames of the --size 2048 workload, timed like
    the headline (synchronise, wall clock, synchronise)."""
    from rescan_line_sted_amd import _lib
    obj, psf, brightness, name = workload(size)
    plan = _lib.DeconvPlan(psf, B, size, size, dtype='f32', device=device)
    plan.set_object(np.broadcast_to(obj, (B, size, size)), brightness)
    for w in range(warmup):
        plan.bench_cycles(K_ITERS, 1, seed=w)
    plan.ctx.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        plan.bench_cycles(K_ITERS, 1, seed=warmup + s)
    plan.ctx.synchronize()
    el = time.perf_counter() - t0
    est = plan.estimate()
    assert np.isfinite(est).all() and est.min() >= 0
    value = B * steps / el
    alg = algorithmic_bytes_per_frame(size * size, len(psf), K_ITERS)
    info = plan.info()
    return {'metric': 'simulated frames/s (%dx%d, 20 RL iters)' % (size, size), 'value': value, 'unit': 'frames/s',
            'steps': steps, 'warmup': warmup, 'ms_per_step': el / steps * 1e3, 'dtype': 'f32',
            'config': {'workload': name, 'frames_per_gpu_per_step': B, 'n_psf': len(psf), 'rl_iters': K_ITERS,
                       'fft': '%dx%d' % (info['ly'], info['lx'])},
            'whole_path': {'algorithmic_bytes_per_frame': alg, 'GBps': alg * value / 1e9, 'frac': alg * value / 1e9 / HBM_PEAK_GBS}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=None, help='frames per GPU per step (default 256; 32 at --size 2048)')
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--size', type=int, default=512, choices=(512, 2048))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-accuracy', action='store_true')
    ap.add_argument('--no-2048', action='store_true', help='skip the short 2048 x 2048 leg of the default run')
    args = ap.parse_args()

    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))

    # Exactly ONE line goes to stdout.  Libraries loaded below (RCCL prints a version banner
    # there) get stderr instead: the process-level stdout is parked and restored for the JSON.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started %d ranks' % (args.gpus, world))
    stub = os.environ.get('RLSTED_BENCH_STUB') == '1'
    size = args.size
    B = args.batch or (256 if size == 512 else 32)
    obj, psf, brightness, workload_name = workload(size)

    # CPU legs first: this process has not touched the GPU yet, the workers are plain children
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not stub:
        cpu = cpu_baseline(size, 10.0 if size == 512 else 20.0)

    comm, transport = (None, None)
    if world > 1 or 'RANK' in os.environ:
        comm, transport = make_comm(rank, world, local_rank, stub)

    if stub:
        plan = StubPlan(psf, B, size, size)
    else:
        from rescan_line_sted_amd import _lib
        plan = _lib.DeconvPlan(psf, B, size, size, dtype=args.dtype, device=local_rank)
    # every frame: same object, its own noise seed (frame index enters the Philox counter)
    plan.set_object(np.broadcast_to(obj, (B, size, size)), brightness)

    def barrier():
        plan.ctx.synchronize()
        if comm is not None:
            comm.barrier()          # device-synchronise, then meet the other ranks

    for w in range(args.warmup):
        plan.bench_cycles(K_ITERS, 1, seed=1000 * rank + w)
    barrier()
    t0 = time.perf_counter()
    # the K steps are enqueued back to back (step s draws with seed ... + s); the device is synchronised on
    # both sides of the timed region, not between steps
    dev_ms = plan.bench_cycles(K_ITERS, args.steps, seed=1000 * rank + args.warmup)
    barrier()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.allreduce_max(elapsed)

    frames_total = B * args.steps * world
    value = frames_total / elapsed

    # the one collective of the path: gather the final estimates on rank 0 over RCCL/xGMI,
    # straight from the plans' device buffers (after the timed region)
    gather = None
    if comm is not None:
        try:
            comm.barrier()
            tg = time.perf_counter()
            _, n_el, _ = comm.gather_plan(plan, [B] * world, 'estimate', 0, to_host=False)
            gather = {'ms': (time.perf_counter() - tg) * 1e3, 'bytes_per_rank': B * size * size * (4 if args.dtype == 'f32' else 8),
                      'frames_on_root': int(n_el // (size * size)) if rank == 0 else None, 'transport': transport}
        except Exception as exc:      # the gather is reported, never allowed to void the measurement
            gather = {'error': repr(exc), 'transport': transport}

    # sanity of what was just computed (not timed)
    est = plan.estimate()
    assert np.isfinite(est).all() and est.min() >= 0

    # ---- roofline of the dominant kernel: in-situ HIP-event durations of one whole cycle (every launch
    # bracketed by events on the stream it goes to, slices overlapping as in the timed steps)
    kt, FL = plan.time_cycle(K_ITERS, seed=4242)
    n_pix, V = size * size, len(psf)
    info = plan.info()
    es = 4 if args.dtype == 'f32' else 8
    avg = {k: v[0] for k, v in kt.items()}
    if 'rl_fused' in avg:        # one launch runs all iterations of all frames
        dom = 'rl_fused'
        launch_bytes = 4 * n_pix * (3 * V + 4) * K_ITERS * FL
        achieved = launch_bytes / (avg[dom] * 1e-3) / 1e9
        iter_ms = avg[dom] / K_ITERS
        alg_iter = launch_bytes / K_ITERS
    else:
        total_ms = {k: kt[k][0] * kt[k][1] for k in ('colconv_H', 'rowpass_RATIO', 'colconv_Ht', 'rowpass_UPDATE')}
        kernel_ms = {'colconv': total_ms['colconv_H'] + total_ms['colconv_Ht'], 'rowpass_RATIO': total_ms['rowpass_RATIO'],
                     'rowpass_UPDATE': total_ms['rowpass_UPDATE']}
        dom = max(kernel_ms, key=kernel_ms.get)
        iter_ms = avg['colconv_H'] + avg['rowpass_RATIO'] + avg['colconv_Ht'] + avg['rowpass_UPDATE']
        alg_iter = 4 * n_pix * (3 * V + 4) * FL     # algorithmic bytes of one RL iteration over one slice
        # Algorithmic bytes exist per PASS: pass 1 (H + ratio) moves 4N(2V+1), pass 2 (H_t +
        # update) 4N(V+3); each pass is one column launch + one row launch and cannot be split between them.
        # achieved = pass bytes / (the pass's two launches); for k_colconv, launched once in each pass, the
        # average over its two passes.
        if dom == 'colconv':
            launch_bytes, pass_ms = alg_iter / 2, iter_ms / 2
        elif dom == 'rowpass_RATIO':
            launch_bytes, pass_ms = 4 * n_pix * (2 * V + 1) * FL, avg['colconv_H'] + avg['rowpass_RATIO']
        else:
            launch_bytes, pass_ms = 4 * n_pix * (V + 3) * FL, avg['colconv_Ht'] + avg['rowpass_UPDATE']
        achieved = launch_bytes / (pass_ms * 1e-3) / 1e9
    # fabric bytes per launch of the dominant kernel from the PMC counters (FETCH_SIZE / WRITE_SIZE,
    # separate rocprofv3 passes, gfx950 correction applied): measured offline with the command recorded
    # in the file, valid for this exact launch shape only.
    traffic = None
    for rnd in ('r02', 'r01'):
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', rnd, 'pmc_traffic.json')))
        except (OSError, ValueError):
            continue
        if (pmc.get('frames_per_launch') == FL and pmc.get('dtype') == args.dtype and pmc.get('n_psf') == V
                and pmc.get('shape') == [size, size] and dom in pmc):
            traffic = pmc[dom].get('fabric_bytes_per_launch', pmc[dom].get('hbm_bytes_per_launch'))
            break
    alg_frame = algorithmic_bytes_per_frame(n_pix, V, K_ITERS)
    # The batch slices run on two streams: on average `concurrency` kernels are in flight, each with its
    # share of the chip, so a kernel's own duration is longer than its cost to the step.
    busy_ms = sum(v[0] * v[1] for v in kt.values())
    concurrency = max(1.0, busy_ms / max(dev_ms / args.steps, 1e-9))
    roofline = {
        'bound': 'hbm', 'kernel': dom, 'unit': 'GB/s', 'peak': HBM_PEAK_GBS,
        'achieved': achieved, 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
        'kernels_in_flight': concurrency, 'frac_per_chip_share': achieved * concurrency / HBM_PEAK_GBS,
        'algorithmic_bytes_per_launch': launch_bytes,
        'timing': 'kernel begin/end HIP events (hipExtLaunchKernelGGL) on every launch of one whole cycle, slices overlapping on their streams as in the timed steps (rl_deconv_time_cycle)',
        'kernel_avg_ms': avg, 'kernel_launches_per_cycle': {k: v[1] for k, v in kt.items()},
        'frames_per_launch': FL,
        'rl_iteration': {'ms': iter_ms, 'algorithmic_GBps': alg_iter / (iter_ms * 1e-3) / 1e9,
                         'frac': alg_iter / (iter_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        'whole_path': {'algorithmic_bytes_per_frame': alg_frame, 'GBps': alg_frame * (value / world) / 1e9,
                       'frac': alg_frame * (value / world) / 1e9 / HBM_PEAK_GBS},
    }

    out = {
        'metric': 'simulated frames/s (%dx%d, 20 RL iters)' % (size, size),
        'value': value, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': workload_name, 'frames_per_gpu_per_step': B, 'n_psf': V, 'rl_iters': K_ITERS,
                   'fft': '%dx%d' % (info['ly'], info['lx']), 'sharding': 'frames over ranks, no data-path collective'},
        'device_ms_per_step': dev_ms / args.steps,
        'final_gather': gather,
        'roofline': roofline,
    }
    if rank == 0 and not args.no_accuracy and not stub and size == 512:   # (the oracle needs minutes per 2048^2 frame)
        out['accuracy'] = accuracy(plan, psf, size, args.dtype)
    if cpu is not None:
        out['cpu_baseline'] = cpu
    # the north star's second reporting size in the same record: a few steps of BASELINE config 3's shape
    # (2048 x 2048, line-rescan, 4 views); single-process runs only -- the N-rank runs measure the headline
    if size == 512 and world == 1 and comm is None and not stub and not args.no_2048 and args.dtype == 'f32':
        try:
            del plan
            out['size_2048'] = second_size(2048, 32, 3, 1, local_rank)
        except Exception as exc:     # reported, never allowed to void the headline
            out['size_2048'] = {'error': repr(exc)}
    if comm is not None:
        assert out['n_gpus'] == args.gpus
        if gather and rank == 0 and 'frames_on_root' in gather:
            assert gather['frames_on_root'] == world * B, gather
        comm.close()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
