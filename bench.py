#!/usr/bin/env python3
"""Headline benchmark: simulated frames/s, 512x512, 20 Richardson-Lucy iterations.

One "step" = one pass of the hot path over one batch of synthetic frames:
for every frame  noiseless = H(object); noisy = Poisson(noiseless) + 1e-9;
estimate = 1; 20 x { estimate *= H_t(noisy / H(estimate)) }   -- i.e.
Deconvolver.create_data_from_object + 20 x Deconvolver.iterate of the reference
(figure_generation/line_sted_tools.py:496-531).  Objects and PSF spectra are
resident in HBM when the timed region starts; nothing crosses PCIe inside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--dtype f32|f64] [--size 512|2048]
                    [--workload headline|fig2sweep]

--gpus N > 1 without a launcher: this process starts `python -m torch.distributed.run
--nproc-per-node N` on itself (before anything touches the GPU) and relays the ranks' one JSON
line.  Under a launcher (RANK set) every rank runs B frames (weak scaling), there is no collective
in the data path, the ranks meet in a barrier before and after the K timed steps, the maximum time
over ranks counts, and ONE gather of the final estimates over RCCL / xGMI follows, timed separately.
The barrier, the reduction and the gather are the C ABI's (rl_comm_*, rl_gather*: RCCL loaded by
librlsted.so itself); torch.distributed is only the fallback transport should that fail to start.

--size 2048 is BASELINE config 3 (synthetic 2048x2048 object, line-rescan, 4 views); the default single-GPU run
also carries short legs of it ("size_2048"), of config 2's other half (512x512 line-rescan, 4 views: "line_rescan_512")
and of the reference's own arithmetic (512x512 point, float64 plan: "f64_512") in the same JSON line (--no-extra-legs
skips them).  With N > 1 ranks (or --workload fig2sweep) the line also carries "fig2_sweep": BASELINE config 4 -- 4 test
objects x 6 doses x 3 scan modes x 16 seeds = 1152 tasks, K = 20 -- cut into cost-weighted shards by
sharding.partition, each rank running its shard through rl_batch_run, ONE gather of the estimates on rank 0
(rl_comm_gather_host), the gather timed separately.
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
PROFILE_ROUND = 'r04'      # profiles/<round>/pmc_traffic*.json: the recorded fabric traffic the roofline objects quote
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ERROR_DEFINITION = ('normwise: max|a-b| / max|b| over the frame; pixelwise: max over pixels with b > 1e-3 max(b) of '
                    '|a-b| / b; a = device estimate, b = float64 oracle on the device-drawn noisy measurement')


def workload(size, n_views=None):
    """512: BASELINE config[1] restricted to the metric's quoted case -- astronaut 128x128 -> 512x512
    (np.kron x4), point-descan STED PSF of the 2.0x operating point (107x107,
    line_sted_figure_2.py:107-120,235-238), brightness 5e10*16.
    2048: config[2] -- default_rng(1234) uniform [0,255) object, the 4 line-rescan views of the same
    operating point, brightness 5e10*256 (the same ~1e7 counts per pixel)."""
    psfs = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
    if size == 512 and (n_views or 1) > 1:      # config 2's other half: astronaut, line-rescan, 4 orientations
        objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
        obj = np.kron(objs['astronaut'].astype(np.float64), np.ones((1, 4, 4)))[0]
        psf = [p[None] for p in psfs['2p0x_lr/line_sted_psfs'][:, 0]]
        return obj, psf, 5e10 * 16, ('astronaut 128->512x512 (np.kron x4), line-rescan STED, 4 views 107x107 (2.0x operating point), '
                                     'simulate (H + Philox Poisson) + 20 RL iterations per frame (BASELINE config 2, line-rescan half)')
    if size == 512:
        objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
        obj = np.kron(objs['astronaut'].astype(np.float64), np.ones((1, 4, 4)))[0]
        psf = [psfs['2p0x_lr/point_sted_psf'][0]]
        name = ('astronaut 128->512x512 (np.kron x4), point-descan STED PSF 107x107 (2.0x operating point), '
                'simulate (H + Philox Poisson) + 20 RL iterations per frame')
        return obj, psf, 5e10 * 16, name
    if (n_views or 4) == 1:     # config 5's tile: one large white-noise frame, point-descan PSF
        obj = np.random.default_rng(4321).random((size, size)) * 255.0
        return obj, [psfs['2p0x_lr/point_sted_psf'][0]], 5e10 * (size // 128) ** 2, (
            'default_rng(4321) uniform %dx%d object, point-descan STED PSF 107x107 (2.0x operating point), simulate (H + Philox '
            'Poisson) + RL iterations per frame (BASELINE config 5 tile)' % (size, size))
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


def _cgroup_cpu_max():
    try:
        return open('/sys/fs/cgroup/cpu.max').read().strip()
    except OSError:
        return 'n/a'


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
    if cores <= 1:
        out['all_cores'] = {'skipped': 'usable cores = 1 (sched_getaffinity %d, os.cpu_count %s, cgroup cpu.max %s): nothing to add to the one-core figure'
                                       % (len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else -1, os.cpu_count(), _cgroup_cpu_max()),
                            'os_cpu_count': os.cpu_count(), 'cpu_model': _cpu_model()}
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

    def gather(self, local, counts, root=0):
        """Host arrays, counts[r] items from rank r (sharding.RcclComm.gather's contract)."""
        local = np.ascontiguousarray(local, dtype=np.float64)
        parts = [None] * self.world if self.rank == root else None
        self.dist.gather_object(local, parts, dst=root)
        if self.rank != root:
            return None
        parts = [p for p, c in zip(parts, counts) if c > 0]
        return np.concatenate(parts) if parts else np.zeros((0,))

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


def make_comm(rank, world, local_rank, stub, allow_torch=False):
    """The product's transport is RCCL through the C ABI (rl_comm_*).  Should it fail to start the run FAILS -- a line measured
    on another transport is not this product's -- unless --allow-torch-transport asks for torch.distributed explicitly."""
    if stub:
        return TorchComm(local_rank, backend='gloo'), 'torch.distributed gloo (stub)'
    from rescan_line_sted_amd import sharding
    try:
        return sharding.RcclComm(rank, world, device=local_rank), 'rl_comm (RCCL through the C ABI)'
    except Exception as exc:
        if not allow_torch:
            raise SystemExit('bench.py: the RCCL communicator of the C ABI did not start (%r); --allow-torch-transport runs on '
                             'torch.distributed instead' % (exc,))
        return TorchComm(local_rank), 'torch.distributed nccl (--allow-torch-transport; rl_comm failed: %r)' % (exc,)


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
            'contract': 1e-5 if dtype == 'f32' else 1e-10,
            'contract_reading': ('BASELINE.json: "<= 1e-5 max rel error vs numpy" is met by the f32 plans in the normwise reading; in the per-pixel reading '
                                 '(pixels above 1e-3 of the maximum) the compliant mode is the float64 plan (1e-8 pixelwise), whose throughput is the '
                                 '"f64_512" leg of this line; an f32 transform of the whole frame cannot do better per pixel (DESIGN.md section 3a; the '
                                 'increment form of H was measured: tools/increment_form_study.py)')}


def extra_leg(size, n_views, dtype, B, steps, warmup, device, k_iters=K_ITERS, comm=None):
    """`steps` whole cycles (simulate + 20 RL iterations) of `B` frames of another workload shape, timed like the
    headline (synchronise, wall clock, synchronise): BASELINE config 3 (2048 x 2048 line-rescan), config 2's line-rescan
    half (512 x 512, 4 views), the headline in the reference's own arithmetic (float64 plan) and config 5's 4096 x 4096
    tile at its 100 iterations."""
    from rescan_line_sted_amd import _lib
    obj, psf, brightness, name = workload(size, n_views)
    plan = _lib.DeconvPlan(psf, B, size, size, dtype=dtype, device=device)
    plan.set_object(np.broadcast_to(obj, (B, size, size)), brightness)
    for w in range(warmup):
        plan.bench_cycles(k_iters, 1, seed=w)
    plan.ctx.synchronize()
    if comm is not None:        # N ranks: replicas of the leg, one per GPU, timed like the headline (barriers, maximum over ranks)
        comm.barrier()
    t0 = time.perf_counter()
    plan.bench_cycles(k_iters, steps, seed=warmup)
    plan.ctx.synchronize()
    if comm is not None:
        comm.barrier()
    el = time.perf_counter() - t0
    world = 1
    if comm is not None:
        el, world = comm.allreduce_max(el), comm.world
    est = plan.estimate()
    assert np.isfinite(est).all() and est.min() >= 0
    value = world * B * steps / el
    es = 4 if dtype == 'f32' else 8
    alg = algorithmic_bytes_per_frame(size * size, len(psf), k_iters) * es // 4
    info = plan.info()
    # fabric bytes / algorithmic bytes of one RL iteration at this leg's launch shape, from the PMC passes recorded for it
    # (profiles/r04: tools/gpu/profile_round.sh; FETCH_SIZE doubled as the microarchitecture guide prescribes)
    fabric = None
    for fn in ('pmc_traffic_%d_%dv_%s.json' % (size, len(psf), dtype), 'pmc_traffic_%d.json' % size):
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', PROFILE_ROUND, fn)))
        except (OSError, ValueError):
            continue
        if pmc.get('n_psf') == len(psf) and pmc.get('dtype') == dtype and pmc.get('shape') == [size, size]:
            fabric = dict(pmc.get('rl_iteration') or {}, frames_per_launch=pmc.get('frames_per_launch'), source='profiles/%s/%s' % (PROFILE_ROUND, fn))
            break
    return {'metric': 'simulated frames/s (%dx%d, %d RL iters)' % (size, size, k_iters), 'value': value, 'unit': 'frames/s',
            'steps': steps, 'warmup': warmup, 'ms_per_step': el / steps * 1e3, 'dtype': dtype,
            'n_gpus': world,
            'config': {'workload': name, 'frames_per_gpu_per_step': B, 'n_psf': len(psf), 'rl_iters': k_iters,
                       'fft': '%dx%d' % (info['ly'], info['lx']), 'frame_pairs': plan.strategy()['frame_pairs']},
            'whole_path': {'algorithmic_bytes_per_frame': alg, 'bytes_per_element': es, 'GBps_per_gpu': alg * value / world / 1e9,
                           'frac': alg * value / world / 1e9 / HBM_PEAK_GBS},
            'rl_iteration_traffic': fabric}


# ------------------------------------------------------------------ BASELINE config 4: the sharded figure-2 sweep
FIG2_DOSES = ('1p0x', '1p5x', '2p0x', '2p5x', '3p0x', '4p0x')
FIG2_SET_NAMES = tuple(d + k for d in FIG2_DOSES for k in ('_point', '_ld', '_lr'))     # the same order on every rank


def fig2_psf_sets(stub):
    """The 18 PSF sets of the sweep: 6 doses x (point-descan, line-descanned, line-rescanned), 1 ... 10 views
    (line_sted_figure_2.py:77-162), made by the product on this rank's device (tune_psf, psf_report, rotations)."""
    doses = FIG2_DOSES
    if stub:      # CPU control-flow tests: shapes only
        views = {'_point': [1] * 6, '_ld': [1, 3, 4, 6, 8, 10], '_lr': [2, 3, 4, 6, 8, 10]}
        return {d + k: [np.ones((1, 7, 7))] * v[i] for k, v in views.items() for i, d in enumerate(doses)}
    from rescan_line_sted_amd import psf
    sets, _ = psf.figure_2_psfs([d + s for d in doses for s in ('_ld', '_lr')])
    out = {}
    for d in doses:
        out[d + '_point'] = [np.asarray(p) for p in sets[d + '_lr_point_sted']]
        for sfx in ('_ld', '_lr'):
            key, = [k for k in sets if k.startswith(d + sfx + '_line_')]
            out[d + sfx] = [np.asarray(p) for p in sets[key]]
    return out


def fig2_psf_sets_once(comm, rank, stub):
    """"PSF sets broadcast once" -- the reference builds them once per figure (line_sted_figure_2.py:66-72,164-165).
    Rank 0 builds the 18 sets; the others receive them through the communicator (two broadcasts: shapes, then values)."""
    if comm is None or not hasattr(comm, 'bcast'):
        return fig2_psf_sets(stub), 'built on every rank (no broadcast in this transport)'
    sets = fig2_psf_sets(stub) if rank == 0 else None
    meta = np.zeros((len(FIG2_SET_NAMES), 3))
    if rank == 0:
        for i, n in enumerate(FIG2_SET_NAMES):
            meta[i] = (len(sets[n]),) + tuple(np.shape(sets[n][0])[-2:])
    meta = comm.bcast(meta, 0).astype(np.int64)
    total = int(sum(v * py * px for v, py, px in meta))
    flat = np.concatenate([np.asarray(p, dtype=np.float64).ravel() for n in FIG2_SET_NAMES for p in sets[n]]) if rank == 0 else np.zeros(total)
    flat = comm.bcast(flat, 0)
    out, o = {}, 0
    for n, (v, py, px) in zip(FIG2_SET_NAMES, meta):
        out[n] = [flat[o + k * py * px:o + (k + 1) * py * px].reshape(1, py, px) for k in range(v)]
        o += v * py * px
    return out, 'built on rank 0, broadcast once (%d values)' % total


def fig2_sweep_leg(comm, rank, world, device, stub):
    """Config 4 end to end: the 1152 tasks dealt to the ranks by plan group (sharding.partition_groups: whole (PSF set, shape)
    groups, a plan's set-up in the cost), this rank's share enqueued group by group (sweep.run_tasks_device: rl_batch_submit on
    cached plans, results in one device buffer), ONE device-to-device gather on rank 0 (rl_comm_gather_device: fp32, unpadded) and
    one download there; run and gather timed separately, maxima over ranks.  No collective in the data path."""
    from rescan_line_sted_amd import sharding, sweep
    objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
    objects = {n: objs[n][0].astype(np.float64) for n in ('astronaut', 'cat', 'lines', 'rings')}
    t0 = time.perf_counter()
    psf_sets, psf_note = fig2_psf_sets_once(comm, rank, stub)
    t_psf = time.perf_counter() - t0
    tasks = sweep.make_tasks(objects, psf_sets, range(16))
    shards, costs = sweep.shard_sweep(tasks, objects, psf_sets, K_ITERS, world)
    keys = sweep.task_groups(tasks, objects)
    mine = [tasks[i] for i in shards[rank]]
    shapes = [tuple(objects[o].shape[-2:]) for o, _, _ in tasks]
    pix = [sum(shapes[i][0] * shapes[i][1] for i in sh) for sh in shards]

    class HostResults:      # the stub's stand-in for sweep.DeviceResults
        def __init__(self, ests):
            self.flat = np.concatenate([e.ravel() for e in ests]) if ests else np.zeros(0)

        def free(self):
            pass

    def run(ts):
        if stub:
            time.sleep(0.001 * len(ts))
            return HostResults([np.ones(objects[o].shape) for o, _, _ in ts])
        return sweep.run_tasks_device(ts, objects, psf_sets, K_ITERS, 5e10, 'f32', device)
    t0 = time.perf_counter()
    run(mine).free()                        # first pass: builds this rank's plans (kept: sweep.plan_for) -- a sweep's set-up
    t_first = time.perf_counter() - t0
    if comm is not None:
        comm.barrier()
    t0 = time.perf_counter()
    res = run(mine)                         # (synchronised inside: every context of the sweep)
    t_run = time.perf_counter() - t0
    t_gather, transport, gbytes = 0.0, 'none (one rank)', int(sum(pix) * 4)
    if comm is not None:
        t_run = comm.allreduce_max(t_run)
        t_first = comm.allreduce_max(t_first)
        comm.barrier()
        t0 = time.perf_counter()
        if not stub and hasattr(comm, 'gather_device'):
            flat = comm.gather_device(res, pix, 0)
            transport = 'rl_comm_gather_device: device buffers, fp32, unpadded, one download on the root'
        else:
            flat = comm.gather(res.flat if stub else np.concatenate([e.ravel() for e in res.download()]), pix, 0)
            transport, gbytes = 'host arrays, float64 (stand-in transport)', int(sum(pix) * 8)
        t_gather = time.perf_counter() - t0
    else:
        flat = res.flat if stub else np.concatenate([e.ravel() for e in res.download()])
    res.free()
    out = {'workload': 'BASELINE config 4: 4 test objects x 6 doses x 3 scan modes x 16 seeds, simulate + %d RL iterations, f32' % K_ITERS,
           'tasks': len(tasks), 'seconds_run_max_over_ranks': t_run, 'frames_per_s': len(tasks) / t_run,
           'seconds_first_pass_with_plan_setup': t_first, 'gather_ms': t_gather * 1e3, 'gather_bytes': gbytes, 'gather_transport': transport,
           'psf_sets_seconds': t_psf, 'psf_sets': psf_note,
           'partition': 'sharding.partition_groups: whole (PSF set, shape) plan groups per rank, largest first; cost = pixels x views x '
                        '(2 + 2K) per task + %d frames of set-up per plan' % sweep.PLAN_SETUP_FRAMES}
    out.update(sharding.partition_stats(shards, costs, keys))
    if not stub:     # (after the timed region) predictions H(est) <= 0 this rank's plans met: 0 = the f32 transforms resolve the sweep's data
        out['unresolved_predictions_this_rank'] = sweep.unresolved_total()
    if rank == 0:
        assert flat.shape[0] == sum(pix) and np.isfinite(flat).all()
        out['frames_on_root'] = len([i for sh in shards for i in sh])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=None, help='frames per GPU per step (default 1024; 32 at --size 2048)')
    ap.add_argument('--workload', default='headline', choices=('headline', 'fig2sweep'),
                    help='fig2sweep: also run BASELINE config 4 (the sharded figure-2 sweep; always run when N > 1)')
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--size', type=int, default=512, choices=(512, 2048))
    ap.add_argument('--allow-torch-transport', action='store_true',
                    help='N > 1: fall back to torch.distributed when the RCCL communicator of the C ABI fails to start (default: exit non-zero)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-accuracy', action='store_true')
    ap.add_argument('--no-extra-legs', '--no-2048', dest='no_extra', action='store_true',
                    help='skip the short 2048 x 2048, 512 x 512 line-rescan and float64 legs of the default run')
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
    os.environ.setdefault('RLSTED_DEVICE', str(local_rank))     # PSF generation (psf.py) runs on this rank's own GPU too
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started %d ranks' % (args.gpus, world))
    stub = os.environ.get('RLSTED_BENCH_STUB') == '1'
    size = args.size
    # 1024 frames per step: the driver's 20 timed steps are then > 1 s of device time (256 frames: 0.27 s)
    B = args.batch or (1024 if size == 512 else 32)
    obj, psf, brightness, workload_name = workload(size)

    # CPU legs first: this process has not touched the GPU yet, the workers are plain children
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not stub:
        cpu = cpu_baseline(size, 6.0 if size == 512 else 20.0)

    comm, transport = (None, None)
    if world > 1 or 'RANK' in os.environ:
        comm, transport = make_comm(rank, world, local_rank, stub, args.allow_torch_transport)

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

    # ---- roofline: in-situ HIP-event durations of one whole cycle (every launch bracketed by the kernel's own
    # begin / end events on the stream it goes to, slices overlapping as in the timed steps)
    kt, FL = plan.time_cycle(K_ITERS, seed=4242)
    n_pix, V = size * size, len(psf)
    info = plan.info()
    avg = {k: v[0] for k, v in kt.items()}
    rl_kernels = ('colconv_H', 'rowpass_RATIO', 'colconv_Ht', 'rowpass_UPDATE')
    total_ms = {k: kt[k][0] * kt[k][1] for k in rl_kernels}
    kernel_ms = {'colconv': total_ms['colconv_H'] + total_ms['colconv_Ht'], 'rowpass_RATIO': total_ms['rowpass_RATIO'],
                 'rowpass_UPDATE': total_ms['rowpass_UPDATE']}
    dom = max(kernel_ms, key=kernel_ms.get)
    # Algorithmic bytes exist per RL ITERATION: 4N(3V+4) per frame -- pass 1 (H + ratio) 4N(2V+1), pass 2
    # (H_t + update) 4N(V+3); each pass is one column launch + one row launch and cannot be split between its two
    # kernels.  So the roofline is quoted on the unit the bytes are defined for: one RL iteration over one slice of
    # the batch = the four launches colconv, ROW_RATIO, colconv, ROW_UPDATE; `achieved` = its algorithmic bytes / the sum
    # of the four launches' average durations, `traffic` = the sum of the four launches' fabric bytes from the PMC
    # counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes, gfx950 correction applied: tools/pmc_traffic.py,
    # recorded offline for this exact launch shape).  The dominant kernel's own share is in `dominant_kernel`.
    iter_ms = sum(avg[k] for k in rl_kernels)
    alg_iter = 4 * n_pix * (3 * V + 4) * FL
    achieved = alg_iter / (iter_ms * 1e-3) / 1e9
    traffic, per_kernel_traffic, pmc_file = None, None, None
    for name in ('pmc_traffic.json', 'pmc_traffic_%d.json' % size):
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', PROFILE_ROUND, name)))
        except (OSError, ValueError):
            continue
        if (pmc.get('frames_per_launch') == FL and pmc.get('dtype') == args.dtype and pmc.get('n_psf') == V
                and pmc.get('shape') == [size, size] and all(k in pmc for k in rl_kernels)):
            per_kernel_traffic = {k: pmc[k]['fabric_bytes_per_launch'] for k in rl_kernels}
            traffic = sum(per_kernel_traffic.values())
            pmc_file = 'profiles/%s/%s' % (PROFILE_ROUND, name)
            break
    alg_frame = algorithmic_bytes_per_frame(n_pix, V, K_ITERS)
    # The batch slices run on two streams: on average `concurrency` kernels are in flight, each with its
    # share of the chip, so a kernel's own duration is longer than its cost to the step.
    busy_ms = sum(v[0] * v[1] for v in kt.values())
    concurrency = max(1.0, busy_ms / max(dev_ms / args.steps, 1e-9))
    roofline = {
        'bound': 'hbm', 'unit': 'GB/s', 'peak': HBM_PEAK_GBS,
        'note': ('priced against the HBM roofline as BASELINE.json asks.  Round 4 (tools/valu_probe.hip with in-kernel cycle stamps, profiles/r04/'
                 'sq_counters_512.txt): a wave issues one vector instruction per 8 cycles at most, a SIMD one per ~2 with >= 4-6 waves issuing; the three '
                 'RL kernels keep the vector pipe ~45 % busy alone and the fabric at 2.8-4.3 TB/s -- neither unit is saturated, the launches are bound by '
                 'how the load / transform / store phases of 6-8 waves per SIMD interleave; two slices in flight reach 4.9-5.1 TB/s of fabric traffic, '
                 '1.75x the algorithmic bytes because every spectrum crosses memory between its row and its column pass (DESIGN.md section 4)') if size < 1024 else
                ('priced against the HBM roofline as BASELINE.json asks; at this size the slices stream through HBM and every kernel of '
                 'the iteration runs at 3.1-4.4 TB/s of fabric traffic alone (profiles/r03/split_2048, point_2048_head): a plain copy on this part '
                 'moves 5.1 TB/s (tools/gpu/gpu_stream_rate.py): the loop is near the streaming rate for its bytes, which are ~2-3x the image-sized passes because every spectrum crosses '
                 'memory between its row and its column pass (DESIGN.md section 3)'),
        'kernel': ('one RL iteration over one slice = 6 launches: split column pass of H (k_colconv_outer<FWD>, <INV>), k_rowpass<RATIO>, '
                   'split column pass of H_t (k_colconv_outer<FWD>, <INV_SUM>), k_rowpass<UPDATE>; colconv_H / colconv_Ht = both halves'
                   if (not stub and plan.strategy().get('split_column_pass')) else
                   'one RL iteration over one slice = 4 launches: k_colconv, k_rowpair<RATIO>, k_colconv, k_rowpair<UPDATE>'),
        'achieved': achieved, 'frac': achieved / HBM_PEAK_GBS,
        'algorithmic_bytes': alg_iter, 'traffic': traffic, 'traffic_over_algorithmic': traffic / alg_iter if traffic else None,
        'traffic_per_kernel': per_kernel_traffic, 'traffic_source': pmc_file,
        'per': 'RL iteration over %d frames (the slice every launch of the loop covers)' % FL,
        'dominant_kernel': {'name': dom, 'share_of_rl_kernel_time': kernel_ms[dom] / sum(kernel_ms.values())},
        'kernels_in_flight': concurrency, 'frac_per_chip_share': achieved * concurrency / HBM_PEAK_GBS,
        'timing': 'kernel begin/end HIP events (hipExtLaunchKernelGGL) on every launch of one whole cycle, slices overlapping on their streams as in the timed steps (rl_deconv_time_cycle)',
        'kernel_avg_ms': avg, 'kernel_launches_per_cycle': {k: v[1] for k, v in kt.items()},
        'frames_per_launch': FL, 'frame_pairs': plan.strategy()['frame_pairs'] if not stub else None,
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
        # predictions H(est) <= 0 met by the ratio kernels in everything this plan ran (rl_deconv_unresolved): 0 = the f32 transforms resolve this workload
        out['accuracy']['unresolved_predictions'] = plan.unresolved()
    if cpu is not None:
        out['cpu_baseline'] = cpu
    # further shapes in the same record, a few steps each (single-process runs only -- the N-rank runs measure the
    # headline): BASELINE config 3 (2048^2 line-rescan), config 2's line-rescan half, the reference's float64 arithmetic
    if size == 512 and world == 1 and comm is None and not stub and not args.no_extra and args.dtype == 'f32':
        del plan
        # (every leg >= 0.5 s of timed device work)
        for key, leg, k in (('size_2048', (2048, 4, 'f32', 32, 5, 1), K_ITERS), ('line_rescan_512', (512, 4, 'f32', 256, 12, 1), K_ITERS),
                            ('f64_512', (512, 1, 'f64', 256, 20, 1), K_ITERS), ('point_2048', (2048, 1, 'f32', 32, 16, 1), K_ITERS),
                            ('f64_2048', (2048, 1, 'f64', 16, 10, 1), K_ITERS), ('size_4096_k100', (4096, 1, 'f32', 8, 3, 1), 100)):
            try:
                out[key] = extra_leg(*leg, local_rank, k)
            except Exception as exc:     # reported, never allowed to void the headline
                out[key] = {'error': repr(exc)}
    # N ranks: the 2048 x 2048 shapes as replicas, one per GPU (north star: "throughput on synthetic 512x512 and 2048x2048 images
    # reported at 1, 2, 4 and 8 GPUs"), timed like the headline
    if world > 1 and size == 512 and not stub and not args.no_extra and args.dtype == 'f32':
        del plan
        for key, leg in (('size_2048', (2048, 4, 'f32', 32, 5, 1)), ('point_2048', (2048, 1, 'f32', 32, 16, 1))):
            try:
                out[key] = extra_leg(*leg, local_rank, K_ITERS, comm)
            except Exception as exc:
                out[key] = {'error': repr(exc)}
    # BASELINE config 4 through the sharder: whenever there is more than one rank, or on request
    if world > 1 or args.workload == 'fig2sweep':
        try:
            out['fig2_sweep'] = fig2_sweep_leg(comm, rank, world, local_rank, stub)
        except Exception as exc:
            out['fig2_sweep'] = {'error': repr(exc)}
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
