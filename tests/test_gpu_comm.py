"""Multi-GPU plumbing and the batch entry points of the C ABI on one MI355X: the RCCL communicator at
world size 1 (a world of 2 cannot share one GPU: RCCL refuses duplicate devices; the N > 1 control flow
is covered on CPU in test_sharding.py / test_bench_cli.py), rl_batch_run, the in-situ cycle timing, the
and launches of more than 65535 images."""
import os

import numpy as np
import pytest

from conftest import max_rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lib():
    from rescan_line_sted_amd import _lib
    assert _lib.device_count() >= 1, 'no GPU visible'
    return _lib


@pytest.fixture(scope='module')
def small(golden):
    psf = [golden('g8_fig2_psfs')['1p5x_lr/point_sted_psf'][0]]
    obj = golden('objects')['rings'].astype(np.float64)[0]
    return psf, obj


def test_rccl_comm_world_1_gather(lib, small, tmp_path):
    from rescan_line_sted_amd import sharding
    psf, obj = small
    comm = sharding.RcclComm(0, 1, device=0, path=str(tmp_path / 'id'))
    assert (comm.rank, comm.world) == (0, 1) and not os.path.exists(str(tmp_path / 'id'))
    assert comm.allreduce_max(3.5) == 3.5
    comm.barrier()
    plan = lib.DeconvPlan(psf, 3, 128, 128, dtype='f32')
    plan.set_object(np.stack([obj, 2 * obj, 3 * obj]), 5e10)
    plan.simulate(seed=3)
    plan.iterate(3)
    got = comm.gather_plan(plan, [2], 'estimate')
    assert np.array_equal(got, plan.estimate()[:2])
    meas = comm.gather_plan(plan, [3], 'measurement')
    assert np.array_equal(meas, plan.measurement())
    ptr, n, dt = comm.gather_plan(plan, [3], 'estimate', to_host=False)
    assert ptr and n == 3 * 128 * 128 and dt == lib.RL_F32
    x = np.arange(24, dtype=np.float64).reshape(4, 3, 2)
    assert np.array_equal(comm.gather(x, [4]), x)
    res = sharding.run_sharded([5, 6, 7], [1.0, 2.0, 3.0], lambda ts: np.array([[t, 2.0 * t] for t in ts]), comm)
    assert np.array_equal(res, [[5, 10], [6, 12], [7, 14]])
    comm.close()


def test_batch_run_equals_the_calls_it_is_made_of(lib, small):
    psf, obj = small
    objs = np.stack([obj * (1 + 0.1 * i) for i in range(5)])
    seeds, ids = np.array([11, 12, 13, 14, 15], dtype=np.uint64), np.array([0, 1, 2, 3, 4], dtype=np.uint32)
    plan = lib.DeconvPlan(psf, 2, 128, 128, dtype='f32')
    est = plan.batch_run(objs, 5e10, seeds, ids, 4)
    assert est.shape == (5, 128, 128)
    single = lib.DeconvPlan(psf, 1, 128, 128, dtype='f32')
    for i in range(5):      # a task's result depends on the task only, not on its chunk
        single.set_object(objs[i], 5e10)
        single.simulate_keyed(seeds[i], ids[i])
        single.reset_estimate()
        single.iterate(4)
        assert np.array_equal(single.estimate()[0], est[i]), i


def test_batch_submit_is_batch_run_without_the_host_in_between(lib, small):
    """rl_batch_submit: the same cycle per task, enqueued -- objects staged in the plan's page-locked blocks (two chunks deep),
    estimates written to caller-owned device memory in the type asked for; rl_batch_run is a submit plus one download.
    Several submits in a row, on two plans of two contexts, then ONE synchronisation each."""
    import ctypes
    from rescan_line_sted_amd import sweep
    psf, obj = small
    objs = [obj * (1 + 0.1 * i) for i in range(7)]
    seeds, ids = np.arange(21, 28, dtype=np.uint64), np.arange(7, dtype=np.uint32)
    ref_plan = lib.DeconvPlan(psf, 2, 128, 128, dtype='f32')
    ref = ref_plan.batch_run(np.stack(objs), 5e10, seeds, ids, 4)         # 4 chunks of 2 (the last one half empty)
    for out_dtype in ('f32', 'f64'):
        res = sweep.DeviceResults([(128, 128)] * 7, out_dtype)
        a = lib.DeconvPlan(psf, 2, 128, 128, dtype='f32')
        b = lib.DeconvPlan(psf, 3, 128, 128, dtype='f32', stream=1)            # another context: its work overlaps a's
        a.batch_submit(objs[:4], 5e10, seeds[:4], ids[:4], 4, res.address(0), out_dtype)
        b.batch_submit(objs[4:], 5e10, seeds[4:], ids[4:], 4, res.address(4), out_dtype)
        a.ctx.synchronize()
        b.ctx.synchronize()
        got = res.download()
        res.free()
        for i in range(7):      # frame pairs: a task's partner differs between the chunkings -- f32 rounding level
            assert max_rel(got[i], ref[i]) < 2e-6, (out_dtype, i)
        # the same chunking: bit for bit
        res = sweep.DeviceResults([(128, 128)] * 7, out_dtype)
        a.batch_submit(objs, 5e10, seeds, ids, 4, res.address(0), out_dtype)
        a.ctx.synchronize()
        got = res.download()
        res.free()
        for i in range(7):
            assert np.array_equal(got[i], ref[i]), (out_dtype, i)
    # unscaled objects (total_brightness <= 0): the levels come from the host sums
    res = sweep.DeviceResults([(128, 128)] * 2, 'f32')
    a = lib.DeconvPlan(psf, 2, 128, 128, dtype='f32')
    a.batch_submit([obj * 1e4, obj * 1.1e4], None, seeds[:2], ids[:2], 3, res.address(0), 'f32')
    a.ctx.synchronize()
    got = res.download()
    a.set_object(np.stack([obj * 1e4, obj * 1.1e4]))
    a.simulate_keyed(seeds[:2], ids[:2])
    a.iterate(3)
    assert np.array_equal(np.stack(got), a.estimate())


def test_sweep_results_gather_and_broadcast_world_1(lib, small, tmp_path):
    """The sweep's transport at world size 1: DeviceResults -> rl_comm_gather_device -> one download; rl_comm_bcast_host."""
    from rescan_line_sted_amd import sharding, sweep
    psf, obj = small
    comm = sharding.RcclComm(0, 1, device=0, path=str(tmp_path / 'id2'))
    objects = {'rings': obj[None], 'wide': np.tile(obj, (1, 1))[None, :96, :]}
    psf_sets = {'a': psf, 'b': [psf[0] * 0.5 + 0.5 * psf[0][:, ::-1, :]]}
    tasks, est = sweep.figure_2_sweep(objects, psf_sets, seeds=(3, 4), iterations=3, comm=comm)
    tasks1, est1 = sweep.figure_2_sweep(objects, psf_sets, seeds=(3, 4), iterations=3, comm=None)
    assert tasks == tasks1 and len(est) == len(tasks) == 8
    for (o, p, s), e, e1 in zip(tasks, est, est1):
        assert e.shape == objects[o].shape[-2:] and np.array_equal(e, e1)
    # every task alone gives the same frame up to the pair partner's rounding
    alone = sweep.run_tasks([tasks[5]], objects, psf_sets, 3)
    assert max_rel(alone[0], est[5]) < 2e-6
    x = np.arange(12, dtype=np.float64).reshape(3, 4)
    assert np.array_equal(comm.bcast(x), x)
    comm.close()
    sweep.clear_plans()


def test_time_cycle_brackets_every_launch(lib, small):
    psf, obj = small
    plan = lib.DeconvPlan(psf, 8, 128, 128, dtype='f32')
    plan.set_object(np.broadcast_to(obj, (8, 128, 128)), 5e10)
    kt, fpl = plan.time_cycle(5, seed=1)
    assert fpl >= 1
    for name in ('colconv_H', 'rowpass_RATIO', 'colconv_Ht', 'rowpass_UPDATE', 'rowpass_FWD', 'rowpass_INV', 'poisson'):
        assert name in kt and kt[name][0] > 0 and kt[name][1] >= 1, (name, kt)
    slices = kt['poisson'][1]
    # H's column pass: once for the simulation, then iterations 2..5 (the first iteration takes H(1) from the plan)
    assert kt['rowpass_RATIO'][1] == 5 * slices and kt['colconv_H'][1] == 5 * slices and kt['rowpass_FWD'][1] == slices
    # the timed cycle leaves what an untimed one leaves
    a = plan.estimate()
    plan.bench_cycles(5, 1, seed=1)
    assert np.array_equal(a, plan.estimate())


def test_time_cycle_counts_a_split_column_pass_once(lib, golden):
    """A multi-view f32 plan on a long column transform: every column pass is two launches (forward half, inverse half);
    the timing API reports them as ONE pass -- count of passes, sum of both launches' durations."""
    psfs = [p[None] for p in golden('g8_fig2_psfs')['2p0x_lr/line_sted_psfs'][:3, 0]]
    B, ny, nx, K = 2, 1200, 96, 3
    plan = lib.DeconvPlan(psfs, B, ny, nx, dtype='f32')
    assert plan.info()['ly'] == 2304 and plan.strategy()['split_column_pass']
    plan.set_object(np.random.default_rng(0).random((B, ny, nx)) * 100, 5e10)
    kt, fpl = plan.time_cycle(K, seed=1)
    slices = kt['poisson'][1]
    assert kt['colconv_H'][1] == K * slices and kt['colconv_Ht'][1] == K * slices     # simulation + iterations 2..K; K x H_t
    assert kt['colconv_H'][0] > 0 and kt['colconv_Ht'][0] > 0
    a = plan.estimate()
    plan.bench_cycles(K, 1, seed=1)
    assert np.array_equal(a, plan.estimate())
    alone = plan.time_kernels(2)          # (works on the plan's buffers: the estimate is not kept)
    assert all(alone[k] > 0 for k in ('colconv_H', 'rowpass_RATIO', 'colconv_Ht', 'rowpass_UPDATE')), alone


def test_more_images_than_grid_y(lib):
    rng = np.random.default_rng(0)
    psf = [rng.random((1, 5, 5))]
    B = 66000
    x = rng.random((B, 12, 12))
    plan = lib.DeconvPlan(psf, B, 12, 12, dtype='f32')
    y = plan.forward(x)
    pick = [0, 1, 65534, 65535, 65536, B - 1]
    ref = lib.DeconvPlan(psf, len(pick), 12, 12, dtype='f32')
    assert np.array_equal(ref.forward(x[pick]), y[pick])
    z = plan.adjoint(y)
    assert np.array_equal(ref.adjoint(y[pick]), z[pick])
