#!/usr/bin/env python3
"""Headline benchmark: simulated frames/s, 512x512, 20 Richardson-Lucy iterations.

One "step" = one pass of the hot path over one batch of synthetic frames:
for every frame  noiseless = H(object); noisy = Poisson(noiseless) + 1e-9;
estimate = 1; 20 x { estimate *= H_t(noisy / H(estimate)) }   -- i.e.
Deconvolver.create_data_from_object + 20 x Deconvolver.iterate of the reference
(figure_generation/line_sted_tools.py:496-531).  Objects and PSF spectra are
resident in HBM when the timed region starts; nothing crosses PCIe inside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--dtype f32|f64]

N > 1: launched by torch.distributed.run, one rank per GPU; frames are sharded
over ranks (weak scaling: B frames per GPU), no collective in the data path,
one gather of the final estimates at the end (outside the timed region it is
reported separately).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NY = NX = 512
K_ITERS = 20
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def workload():
    """BASELINE config[1] restricted to the metric's quoted case: astronaut
    128x128 -> 512x512 (np.kron x4), point-descan STED PSF of the 2.0x operating
    point (107x107, line_sted_figure_2.py:107-120,235-238), brightness 5e10*16."""
    objs = np.load(os.path.join(ROOT, 'tests', 'golden', 'objects.npz'))
    psfs = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8_fig2_psfs.npz'))
    obj = np.kron(objs['astronaut'].astype(np.float64), np.ones((1, 4, 4)))[0]
    psf = [psfs['2p0x_lr/point_sted_psf'][0]]
    return obj, psf, 5e10 * 16


def algorithmic_bytes_per_frame(n_pix, n_psf, k):
    # fp32 storage, each array touched once per logical pass
    return 4 * n_pix * ((2 * n_psf + 2) + k * (3 * n_psf + 4))


def cpu_baseline(obj, psf, brightness, budget_s=12.0):
    """The oracle (numpy float64 restatement of the reference, one thread) on a
    bounded sample of the same workload."""
    from oracle import line_sted_oracle as orc
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    t0 = time.perf_counter()
    frames = 0
    while True:
        d = orc.Deconvolver(psf)
        d.create_data_from_object(obj[None].copy(), brightness, random_seed=frames)
        for _ in range(K_ITERS):
            d.iterate()
        frames += 1
        el = time.perf_counter() - t0
        if el > budget_s or frames >= 64:
            break
    return {'value': frames / el, 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
            'sample': '%d frames of the same 512x512 / K=20 workload, numpy float64 oracle, 1 thread, %.1f s'
                      % (frames, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=256, help='frames per GPU per step')
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--kernel-reps', type=int, default=20)
    args = ap.parse_args()

    # Exactly ONE line goes to stdout.  Libraries loaded below (RCCL prints a version banner
    # there) get stderr instead: the process-level stdout is parked and restored for the JSON.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1 or 'RANK' in os.environ:      # launched by torch.distributed.run (also with one rank)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl')       # RCCL on ROCm
        sync_t = torch.zeros(1, device='cuda')

    from rescan_line_sted_amd import _lib
    obj, psf, brightness = workload()
    B = args.batch
    plan = _lib.DeconvPlan(psf, B, NY, NX, dtype=args.dtype, device=local_rank)
    # every frame: same object, its own noise seed (frame index enters the Philox counter)
    plan.set_object(np.broadcast_to(obj, (B, NY, NX)), brightness)

    def barrier():
        plan.ctx.synchronize()
        if dist is not None:
            dist.all_reduce(sync_t)
            import torch
            torch.cuda.synchronize()

    for w in range(args.warmup):
        plan.bench_cycles(K_ITERS, 1, seed=1000 * rank + w)
    barrier()
    t0 = time.perf_counter()
    dev_ms = 0.0
    for s in range(args.steps):
        dev_ms += plan.bench_cycles(K_ITERS, 1, seed=1000 * rank + args.warmup + s)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    frames_total = B * args.steps * world
    value = frames_total / elapsed

    # the one collective of the path: gather the final estimates on rank 0 over
    # RCCL/xGMI, straight from the plan's device buffer (after the timed region)
    gather = None
    if dist is not None:
        try:
            import torch
            from rescan_line_sted_amd import sharding
            est_dev = torch.as_tensor(plan.device_array('estimate'), device='cuda')
            torch.cuda.synchronize()
            tg = time.perf_counter()
            got = sharding.gather_to_root(est_dev, [B] * world, dist)
            torch.cuda.synchronize()
            gather = {'ms': (time.perf_counter() - tg) * 1e3, 'bytes_per_rank': est_dev.numel() * est_dev.element_size(),
                      'frames_on_root': int(got.shape[0]) if got is not None else None}
            del got
        except Exception as exc:      # the gather is reported, never allowed to void the measurement
            gather = {'error': repr(exc)}

    # sanity of what was just computed (not timed)
    est = plan.estimate()
    assert np.isfinite(est).all() and est.min() >= 0

    # dominant-kernel roofline, measured live with HIP events on the plan's stream
    kt = plan.time_kernels(args.kernel_reps)
    FL = kt.pop('frames_per_rl_launch')         # frames per launch of the RL kernels (batch slices)
    n_pix, V = NY * NX, len(psf)
    info = plan.info()
    es = 4 if args.dtype == 'f32' else 8
    spec = NY * info['pitch'] * 2 * es          # one row-transformed half spectrum
    img = n_pix * es
    launches = {                                # kernel -> (launches per frame-cycle, implementation bytes per frame per launch)
        'colconv_H': (K_ITERS + 1, spec + V * spec),
        'rowpass_RATIO': (K_ITERS, V * (2 * spec + img)),
        'colconv_Ht': (K_ITERS, 2 * V * spec),
        'rowpass_UPDATE': (K_ITERS, V * spec + 3 * img + spec),
    }
    per_cycle_ms = {k: kt[k] * launches[k][0] for k in launches}
    # the two column launches of an iteration are one kernel (k_colconv): judged together
    kernel_ms = {'colconv': per_cycle_ms['colconv_H'] + per_cycle_ms['colconv_Ht'],
                 'rowpass_RATIO': per_cycle_ms['rowpass_RATIO'], 'rowpass_UPDATE': per_cycle_ms['rowpass_UPDATE']}
    dom = max(kernel_ms, key=kernel_ms.get)
    iter_ms = kt['colconv_H'] + kt['rowpass_RATIO'] + kt['colconv_Ht'] + kt['rowpass_UPDATE']
    alg_iter = 4 * n_pix * (3 * V + 4) * FL     # algorithmic bytes of one RL iteration over one slice
    # the dominant kernel's share of the iteration's algorithmic bytes: pass 1
    # (H + ratio) moves 4N(2V+1), pass 2 (H_t + update) 4N(V+3); each pass is one
    # column launch + one row launch, the bytes are attributed to the pass's row
    # kernel (which touches the images) and the column kernel is charged its
    # pass's bytes as well, i.e. achieved = pass bytes / (pass's two launches).
    pass1 = 4 * n_pix * (2 * V + 1) * FL
    pass2 = 4 * n_pix * (V + 3) * FL
    if dom == 'colconv':      # one launch in each pass: half the iteration's bytes per launch pair, on average
        pass_bytes, pass_ms = alg_iter / 2, iter_ms / 2
    elif dom == 'rowpass_RATIO':
        pass_bytes, pass_ms = pass1, kt['colconv_H'] + kt['rowpass_RATIO']
    else:
        pass_bytes, pass_ms = pass2, kt['colconv_Ht'] + kt['rowpass_UPDATE']
    achieved = pass_bytes / (pass_ms * 1e-3) / 1e9
    # HBM bytes per launch of the dominant kernel from the PMC counters (FETCH_SIZE /
    # WRITE_SIZE, separate rocprofv3 passes, gfx950 correction applied): measured offline
    # with the command recorded in the file, valid for this exact launch shape only.
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01', 'pmc_traffic.json')))
        if (pmc.get('frames_per_launch') == FL and pmc.get('dtype') == args.dtype and pmc.get('n_psf') == V
                and pmc.get('shape') == [NY, NX] and dom in pmc):
            traffic = pmc[dom]['hbm_bytes_per_launch']      # 'colconv': the same for the H and the H_t launch
    except (OSError, ValueError):
        pass
    roofline = {
        'bound': 'hbm', 'kernel': dom, 'unit': 'GB/s', 'peak': HBM_PEAK_GBS,
        'achieved': achieved, 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
        'algorithmic_bytes_per_launch_pair': pass_bytes,
        'kernel_avg_ms': kt,
        'frames_per_launch': FL,
        'kernel_moved_GBps': {k: launches[k][1] * FL / (kt[k] * 1e-3) / 1e9 for k in launches},
        'rl_iteration': {'ms': iter_ms, 'algorithmic_GBps': alg_iter / (iter_ms * 1e-3) / 1e9,
                         'frac': alg_iter / (iter_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        'whole_path': {'algorithmic_bytes_per_frame': algorithmic_bytes_per_frame(n_pix, V, K_ITERS),
                       'GBps': algorithmic_bytes_per_frame(n_pix, V, K_ITERS) * (value / world) / 1e9,
                       'frac': algorithmic_bytes_per_frame(n_pix, V, K_ITERS) * (value / world) / 1e9 / HBM_PEAK_GBS},
    }

    out = {
        'metric': 'simulated frames/s (512x512, 20 RL iters)',
        'value': value, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': 'astronaut 128->512x512 (np.kron x4), point-descan STED PSF 107x107 (2.0x operating point), '
                               'simulate (H + Philox Poisson) + 20 RL iterations per frame',
                   'frames_per_gpu_per_step': B, 'n_psf': V, 'rl_iters': K_ITERS,
                   'fft': '%dx%d' % (info['ly'], info['lx']), 'sharding': 'frames over ranks, no data-path collective'},
        'device_ms_per_step': dev_ms / args.steps,
        'final_gather': gather,
        'roofline': roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(obj, psf, brightness)
    if dist is not None:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
