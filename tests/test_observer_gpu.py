"""GPU parity: DiscreteEKFObserver (sofacontrol/tpwl/observer.py:33-126) on the device against the golden
vectors of the imported reference (g9_ekf) and the oracle.  Tolerance 1e-9 relative: the kernel solves with a
Cholesky factor of S where the reference multiplies by inv(S)."""
import io
import contextlib

import numpy as np
import pytest

from oracle import tpwl as otpwl, observer as oobs
from helpers import golden_problem, product_tpwl, meas_selector

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-9):
    np.testing.assert_allclose(a, b, rtol=0, atol=rtol * max(1.0, float(np.abs(b).max())))


def _model(method='nn', beta=None):
    model, U, q_ref, v_ref, Hf = golden_problem(4, 3, 7, 20, 50, q_scale=0.3)
    Cf = meas_selector([3, 9], 20)
    tp = product_tpwl(model, U, q_ref, v_ref, Hf, Cf=Cf, method=method, beta=beta)
    return model, tp, (U, q_ref, v_ref)


def test_ekf_golden(golden):
    from sofacontrol_amd.tpwl.observer import DiscreteEKFObserver
    g = golden('g9_ekf')
    model, tp, _ = _model()
    dt = 0.02
    with contextlib.redirect_stdout(io.StringIO()):
        tp.pre_discretize(dt)
    ekf = DiscreteEKFObserver(tp, Sigma0=g['Sigma0'].copy(), W=g['W'], V=g['V'])
    close(ekf.x, g['x_init'], 1e-12); close(ekf.z, g['z_init'], 1e-12)
    np.testing.assert_array_equal(ekf.Sigma, g['Sigma0'])
    assert ekf.get_meas_dim() == 6
    assert set(ekf.get_observer_params()) == {'W', 'V', 'meas_dim', 'state_dim', 'C', 'H'}
    for k in range(8):
        ekf.update(g['u'][k], g['y'][k], dt)
        close(ekf.x, g['x'][k]); close(ekf.Sigma, g['Sigma'][k]); close(ekf.z, g['z'][k])
    ekf.predict_state(g['u'][0], dt)
    close(ekf.x, g['x_pred']); close(ekf.Sigma, g['Sigma_pred'])
    x = ekf.update_state(g['y'][1])
    close(x, g['x_upd']); close(ekf.Sigma, g['Sigma_upd'])
    ekf.initialize(g['xf0'])
    close(ekf.x, g['x_reinit'], 1e-11)


def test_ekf_requires_measurement_model():
    from sofacontrol_amd.tpwl.observer import DiscreteEKFObserver
    model, U, q_ref, v_ref, Hf = golden_problem(4, 3, 7, 20, 50)
    tp = product_tpwl(model, U, q_ref, v_ref, Hf)
    with pytest.raises(RuntimeError):
        DiscreteEKFObserver(tp)


def test_ekf_diamond_size_vs_oracle():
    """Diamond-sized filter (n_x = 60, n_y = 30) against the oracle on a seeded model."""
    from sofacontrol_amd.tpwl.observer import DiscreteEKFObserver
    r, m, P, nodes = 30, 4, 16, 40
    model, U, q_ref, v_ref, Hf = golden_problem(r, m, P, nodes, 70, q_scale=0.3)
    Cf = meas_selector(list(range(2, 22, 2)), nodes)
    tp = product_tpwl(model, U, q_ref, v_ref, Hf, Cf=Cf)
    dt = 0.01
    with contextlib.redirect_stdout(io.StringIO()):
        tp.pre_discretize(dt)
    rng = np.random.default_rng(5)
    n, ny = 2 * r, 30
    W = 100 * np.eye(n); V = np.eye(ny)
    ekf = DiscreteEKFObserver(tp, W=W, V=V)
    Ad, Bd, dd = np.stack(tp.A_d), np.stack(tp.B_d), np.stack(tp.d_d)
    x, S = np.zeros(n), np.eye(n)
    Cm, y_ref = np.asarray(tp.C), tp.y_ref
    for k in range(5):
        u = rng.uniform(0, 300, m)
        y = y_ref + 0.05 * rng.standard_normal(ny)
        x, S = oobs.predict(model, Ad, Bd, dd, x, S, u, W)
        x, S = oobs.update(Cm, y_ref, x, S, y, V)
        ekf.update(u, y, dt)
        close(ekf.x, x); close(ekf.Sigma, S)


def test_ekf_weighting_model(golden):
    """Weighting-mode model: the predictor takes the blended, host-discretised (A_d, B_d, d_d)."""
    from sofacontrol_amd.tpwl.observer import DiscreteEKFObserver
    g = golden('g9_ekf')
    model, tp, _ = _model(method='weighting', beta=2.0)
    dt = 0.02
    ekf = DiscreteEKFObserver(tp, Sigma0=g['Sigma0'].copy(), W=g['W'], V=g['V'])
    x, S = ekf.x.copy(), g['Sigma0']
    Cm, y_ref = np.asarray(tp.C), tp.y_ref
    for k in range(3):
        A, B, d = otpwl.weighted_jacobians(model, x, 2.0, dt, 'zoh')
        x, S = A @ x + B @ g['u'][k] + d, A @ S @ A.T + g['W']
        x, S = oobs.update(Cm, y_ref, x, S, g['y'][k], g['V'])
        ekf.update(g['u'][k], g['y'][k], dt)
        close(ekf.x, x); close(ekf.Sigma, S)


def test_ekf_diamond_size_weighting_and_partial_steps():
    """MFMA filter kernel with explicit (A_d, B_d, d_d) (weighting-mode model) and predict-only / update-only calls."""
    from sofacontrol_amd.tpwl.observer import DiscreteEKFObserver
    r, m, P, nodes = 30, 4, 8, 40
    model, U, q_ref, v_ref, Hf = golden_problem(r, m, P, nodes, 71, q_scale=0.3)
    Cf = meas_selector(list(range(2, 22, 2)), nodes)
    tp = product_tpwl(model, U, q_ref, v_ref, Hf, Cf=Cf, method='weighting', beta=2.0, discr='fe')
    dt = 0.01
    rng = np.random.default_rng(6)
    n, ny = 2 * r, 30
    W = 50 * np.eye(n); V = 2 * np.eye(ny)
    ekf = DiscreteEKFObserver(tp, W=W, V=V)
    x, S = np.zeros(n), np.eye(n)
    Cm, y_ref = np.asarray(tp.C), tp.y_ref
    for k in range(3):
        u = rng.uniform(0, 300, m)
        y = y_ref + 0.05 * rng.standard_normal(ny)
        A, B, d = otpwl.weighted_jacobians(model, x, 2.0, dt, 'fe')
        x, S = A @ x + B @ u + d, A @ S @ A.T + W
        if k == 1:
            ekf.predict_state(u, dt)
            close(ekf.x, x); close(ekf.Sigma, S)
            x, S = oobs.update(Cm, y_ref, x, S, y, V)
            ekf.update_state(y)
        else:
            x, S = oobs.update(Cm, y_ref, x, S, y, V)
            ekf.update(u, y, dt)
        close(ekf.x, x); close(ekf.Sigma, S)


def test_ekf_step_projected_golden(golden):
    """sekf_step_projected (projection on a side stream beside the filter kernel): the filter half reproduces the g9
    vectors of the imported reference, the projection half equals compute_RO_state bit for bit and the oracle."""
    from sofacontrol_amd.tpwl.observer import DiscreteEKFObserver
    from oracle import pod as opod
    g = golden('g9_ekf')
    model, tp, (U, q_ref, v_ref) = _model()
    dt = 0.02
    with contextlib.redirect_stdout(io.StringIO()):
        tp.pre_discretize(dt)
    ekf = DiscreteEKFObserver(tp, Sigma0=g['Sigma0'].copy(), W=g['W'], V=g['V'])
    rng = np.random.default_rng(11)
    n_f = U.shape[0]
    for k in range(g['u'].shape[0]):
        xf = np.concatenate([v_ref, q_ref]) + rng.standard_normal(2 * n_f)
        xr = ekf.update_projected(tp.rom, xf, g['u'][k], g['y'][k], dt)
        np.testing.assert_array_equal(xr, tp.rom.compute_RO_state(xf=xf))
        want = np.concatenate([opod.project(U, v_ref, xf[None, :n_f])[0], opod.project(U, q_ref, xf[None, n_f:])[0]])
        close(xr, want, 1e-12)
        close(ekf.x, g['x'][k]); close(ekf.Sigma, g['Sigma'][k]); close(ekf.z, g['z'][k])
    with pytest.raises(RuntimeError):
        ekf.update_projected(tp.rom, np.zeros(3), g['u'][0], g['y'][0], dt)


def test_ekf_step_projected_diamond_size():
    """Full Diamond shape (n_f = 4884, r = 30, n_y = 30): fused call against the two separate calls on a twin filter."""
    from sofacontrol_amd.tpwl.observer import DiscreteEKFObserver
    r, m, P, nodes = 30, 4, 16, 1628
    model, U, q_ref, v_ref, Hf = golden_problem(r, m, P, nodes, 72, q_scale=0.3)
    Cf = meas_selector(list(range(2, 22, 2)), nodes)
    tp = product_tpwl(model, U, q_ref, v_ref, Hf, Cf=Cf)
    dt = 0.01
    with contextlib.redirect_stdout(io.StringIO()):
        tp.pre_discretize(dt)
    n, ny = 2 * r, 30
    a = DiscreteEKFObserver(tp, W=100 * np.eye(n), V=np.eye(ny))
    b = DiscreteEKFObserver(tp, W=100 * np.eye(n), V=np.eye(ny))
    rng = np.random.default_rng(8)
    for k in range(4):
        u = rng.uniform(0, 300, m)
        y = tp.y_ref + 0.05 * rng.standard_normal(ny)
        xf = tp.rom.x_ref + rng.standard_normal(2 * U.shape[0])
        xr = a.update_projected(tp.rom, xf, u, y, dt)
        want = tp.rom.compute_RO_state(xf=xf)
        b.update(u, y, dt)
        np.testing.assert_array_equal(xr, want)
        np.testing.assert_array_equal(a.x, b.x)
        np.testing.assert_array_equal(a.Sigma, b.Sigma)


@pytest.mark.parametrize('r,method', [(36, 'nn'), (36, 'weighting'), (33, 'nn'), (35, 'nn')])
def test_ekf_wide_kernel_vs_oracle(r, method):
    """64 < n_x <= 80 (the shipped Diamond basis r = 36 -> n_x = 72, and sizes that are not multiples of 4 / 16): the
    MFMA kernel that reads A^T from the table instead of staging it; against the oracle, incl. predict-only and
    update-only calls and the explicit-Jacobian (weighting) form."""
    from sofacontrol_amd.tpwl.observer import DiscreteEKFObserver
    from sofacontrol_amd import _lib
    m, P, nodes = 4, 8, 44
    model, U, q_ref, v_ref, Hf = golden_problem(r, m, P, nodes, 73 + r, q_scale=0.3)
    Cf = meas_selector(list(range(2, 22, 2)), nodes)
    kw = dict(method='weighting', beta=2.0, discr='fe') if method == 'weighting' else {}
    tp = product_tpwl(model, U, q_ref, v_ref, Hf, Cf=Cf, **kw)
    dt = 0.01
    if method == 'nn':
        with contextlib.redirect_stdout(io.StringIO()):
            tp.pre_discretize(dt)
        Ad, Bd, dd = np.stack(tp.A_d), np.stack(tp.B_d), np.stack(tp.d_d)
    rng = np.random.default_rng(r)
    n, ny = 2 * r, 30
    W = 100 * np.eye(n) + 0.1 * np.diag(rng.uniform(0, 1, n)); V = np.eye(ny) + 0.05 * np.diag(rng.uniform(0, 1, ny))
    ekf = DiscreteEKFObserver(tp, W=W, V=V)
    x, S = np.zeros(n), np.eye(n)
    Cm, y_ref = np.asarray(tp.C), tp.y_ref
    for k in range(5):
        u = rng.uniform(0, 300, m)
        y = y_ref + 0.05 * rng.standard_normal(ny)
        if method == 'nn':
            xp, Sp = oobs.predict(model, Ad, Bd, dd, x, S, u, W)
        else:
            A, B, d = otpwl.weighted_jacobians(model, x, 2.0, dt, 'fe')
            xp, Sp = A @ x + B @ u + d, A @ S @ A.T + W
        if k == 2:                                            # predict only, then update only
            ekf.predict_state(u, dt)
            close(ekf.x, xp); close(ekf.Sigma, Sp)
            ekf.update_state(y)
        else:
            ekf.update(u, y, dt)
        x, S = oobs.update(Cm, y_ref, xp, Sp, y, V)
        close(ekf.x, x); close(ekf.Sigma, S)
