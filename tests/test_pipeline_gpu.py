"""The device-resident chain of one receding-horizon step (round 3): srom_project_dev -> srom_qv2x_dev (utils.qv2x,
utils.py:129-130) -> stpwl_rollout_dev (TPWL.rollout, tpwl.py:193-216) -> sgusto_plan_solve_dev -- each link against the
host-pointer entry point / the oracle, then the whole chain against the same steps through the Python classes."""
import ctypes as C
import io
import contextlib

import numpy as np
import pytest

from oracle import tpwl as otpwl, pod as opod
from helpers import product_tpwl, tip_selector, Poly

pytestmark = pytest.mark.gpu


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def test_qv2x_dev_pitches_and_missing_velocities():
    from sofacontrol_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(0)
    B, r = 37, 5
    q, v = rng.standard_normal((B, r + 3)), rng.standard_normal((B, r + 1))       # pitched sources: the first r columns count
    dq, dv = _lib.DeviceBuffer.from_array(q), _lib.DeviceBuffer.from_array(v)
    out = np.full((B, 2 * r + 2), 7.0)
    dx = _lib.DeviceBuffer.from_array(out)
    _lib.check(L.srom_qv2x_dev(dq.ptr, C.c_int64(r + 3), dv.ptr, C.c_int64(r + 1), C.c_int64(B), C.c_int(r), dx.ptr, C.c_int64(2 * r + 2), None), 'qv2x')
    _lib.sync()
    got = dx.to_array(out.shape)
    assert np.array_equal(got[:, :r], v[:, :r]) and np.array_equal(got[:, r:2 * r], q[:, :r]) and np.all(got[:, 2 * r:] == 7.0)
    _lib.check(L.srom_qv2x_dev(dq.ptr, C.c_int64(r + 3), None, C.c_int64(0), C.c_int64(B), C.c_int(r), dx.ptr, C.c_int64(2 * r + 2), None), 'qv2x')
    _lib.sync()
    got = dx.to_array(out.shape)
    assert np.all(got[:, :r] == 0.0) and np.array_equal(got[:, r:2 * r], q[:, :r])          # utils.qv2x(q, 0)


@pytest.mark.parametrize('r,m,P,N,batch', [(30, 4, 64, 50, 6), (5, 3, 9, 10, 1), (36, 4, 20, 12, 3)])
def test_rollout_dev_equals_host_entry_point_and_oracle(r, m, P, N, batch):
    from sofacontrol_amd import _lib
    L = _lib.lib()
    model = otpwl.synthetic_model(r, m, P, seed=r + m)
    model['q'] *= 0.1
    rng = np.random.default_rng(2)
    U, _ = np.linalg.qr(rng.standard_normal((3 * 40, r)))
    tp = product_tpwl(model, U, np.zeros(120), np.zeros(120), tip_selector(7, 40))
    dt = 0.05
    quiet(tp.pre_discretize, dt)
    x0 = 0.01 * rng.standard_normal((batch, 2 * r))
    u = rng.uniform(0, 800, (batch, N, m))
    Xh, Zh = tp.rollout(x0, u, dt)
    nz = Zh.shape[-1]
    dx0, du = _lib.DeviceBuffer.from_array(x0), _lib.DeviceBuffer.from_array(u)
    dX, dZ = _lib.DeviceBuffer(batch * (N + 1) * 2 * r * 8), _lib.DeviceBuffer(batch * (N + 1) * nz * 8)
    _lib.check(L.stpwl_rollout_dev(tp.handle_for(dt), dx0.ptr, du.ptr, C.c_int(N), C.c_int64(batch), dX.ptr, dZ.ptr, None), 'rollout_dev')
    _lib.sync()
    assert np.array_equal(dX.to_array((batch, N + 1, 2 * r)), Xh) and np.array_equal(dZ.to_array((batch, N + 1, nz)), Zh)
    Ad, Bd, dd = np.stack(tp.A_d), np.stack(tp.B_d), np.stack(tp.d_d)
    xo = otpwl.rollout(model, Ad, Bd, dd, x0[0], u[0])
    np.testing.assert_allclose(Xh[0], xo, rtol=0, atol=1e-10 * max(1.0, np.abs(xo).max()))
    # outputs are optional
    _lib.check(L.stpwl_rollout_dev(tp.handle_for(dt), dx0.ptr, du.ptr, C.c_int(N), C.c_int64(batch), dX.ptr, None, None), 'rollout_dev')
    _lib.sync()
    assert np.array_equal(dX.to_array((batch, N + 1, 2 * r)), Xh)


def test_whole_step_on_the_device_equals_the_class_level_steps():
    """project -> qv2x -> zero-input rollout -> GuSTO on resident buffers == POD.compute_RO_state -> utils.qv2x ->
    TPWL.rollout -> GuSTO.solve through the Python classes (host buffers), bit for bit."""
    from sofacontrol_amd import _lib, utils as scutils
    from sofacontrol_amd.mor.pod import POD
    from sofacontrol_amd.scp.models.tpwl import TPWLGuSTO
    from sofacontrol_amd.scp.gusto import GuSTO
    L = _lib.lib()
    r, m, P, N, dt, B = 6, 3, 9, 10, 0.05, 5
    model = otpwl.synthetic_model(r, m, P, seed=41)
    model['q'] *= 0.2
    rng = np.random.default_rng(4)
    n_nodes = 40
    n_f = 3 * n_nodes
    U, _ = np.linalg.qr(rng.standard_normal((n_f, r)))
    q_ref = rng.uniform(-1, 1, n_f)
    tp = product_tpwl(model, U, q_ref, np.zeros(n_f), tip_selector(7, n_nodes))
    rom = POD(dict(U=U, q_ref=q_ref, v_ref=np.zeros(n_f)))
    gm = TPWLGuSTO(tp)
    quiet(gm.pre_discretize, dt)
    n = 2 * r
    Qz = np.diag([0, 0, 0, 100., 100., 0]); R = 1e-5 * np.eye(m)
    UA = np.kron(np.eye(m), np.array([[1.], [-1.]])); Ub = np.tile([800., 0.], m)
    Xfull = q_ref + 0.05 * rng.standard_normal((B, n_f))
    z = np.tile(0.01 * np.sin(np.linspace(0, 2, N + 1))[:, None] * np.array([0, 0, 0, 1., -0.5, 0]), (B, 1, 1))
    # class level, host buffers
    x0 = scutils.qv2x(rom.compute_RO_state(qf=Xfull), np.zeros((B, r)))
    u_init = np.zeros((B, N, m))
    x_init, _ = tp.rollout(x0, u_init, dt)
    g = quiet(GuSTO, gm, N, dt, Qz, R, x0, u_init, x_init, z=z, U=Poly(UA, Ub), batch=B, max_gusto_iters=5, verbose=0)
    xo, uo, zo, _ = g.get_solution()
    # the same step on resident buffers
    dXf, dq = _lib.DeviceBuffer.from_array(Xfull), _lib.DeviceBuffer(B * r * 8)
    dx0, du0, dxi, dz = _lib.DeviceBuffer(B * n * 8), _lib.DeviceBuffer.from_array(u_init), _lib.DeviceBuffer(B * (N + 1) * n * 8), _lib.DeviceBuffer.from_array(z)
    nz = zo.shape[-1]
    ox, ou, oz = _lib.DeviceBuffer(B * (N + 1) * n * 8), _lib.DeviceBuffer(B * N * m * 8), _lib.DeviceBuffer(B * (N + 1) * nz * 8)
    oi, os_ = _lib.DeviceBuffer(B * 4), _lib.DeviceBuffer(B * 4)
    _lib.check(L.srom_project_dev(rom.handle, 0, dXf.ptr, C.c_int64(B), C.c_int64(n_f), dq.ptr, C.c_int64(r), None), 'project')
    _lib.check(L.srom_qv2x_dev(dq.ptr, C.c_int64(r), None, C.c_int64(0), C.c_int64(B), C.c_int(r), dx0.ptr, C.c_int64(n), None), 'qv2x')
    _lib.check(L.stpwl_rollout_dev(tp.handle_for(dt), dx0.ptr, du0.ptr, C.c_int(N), C.c_int64(B), dxi.ptr, None, None), 'rollout')
    _lib.check(L.sgusto_plan_solve_dev(g.plan, dx0.ptr, du0.ptr, dxi.ptr, dz.ptr, None, None, ox.ptr, ou.ptr, oz.ptr, oi.ptr, os_.ptr, None, None), 'gusto')
    _lib.sync()
    assert np.array_equal(dx0.to_array((B, n)), x0) and np.array_equal(dxi.to_array((B, N + 1, n)), x_init)
    assert np.array_equal(ox.to_array((B, N + 1, n)), np.asarray(xo).reshape(B, N + 1, n))
    assert np.array_equal(ou.to_array((B, N, m)), np.asarray(uo).reshape(B, N, m))
    assert np.all(os_.to_array((B,), dtype=np.int32) == 0)
