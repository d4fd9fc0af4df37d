"""GPU parity: TPWL nearest point / Jacobian gather / rollout / characteristic values against the golden
vectors of the imported reference and the oracle.  Indices bit-exact; float64 values to 1e-11."""
import io
import contextlib

import numpy as np
import pytest

from oracle import tpwl as otpwl
from helpers import golden_problem, product_tpwl

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-11):
    np.testing.assert_allclose(a, b, rtol=0, atol=rtol * max(1.0, float(np.abs(b).max())))


def quiet(fn, *a):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a)


def test_golden_g3(golden):
    g = golden('g3_tpwl')
    model, U, q_ref, v_ref, Hf = golden_problem(4, 3, 7, 20, 10)
    model['w_v'] = 0.5
    tp = product_tpwl(model, U, q_ref, v_ref, Hf)
    close(tp.H, g['H']); close(tp.z_ref, g['z_ref'])
    X = g['X']
    assert [tp.calc_nearest_point(x) for x in X] == list(g['nearest'])
    assert np.array_equal(tp.calc_nearest_point(X), g['nearest'])
    dt = 0.05
    for meth in ('fe', 'be', 'bil', 'zoh'):
        t2 = product_tpwl(model, U, q_ref, v_ref, Hf, discr=meth)
        quiet(t2.pre_discretize, dt)
        close(np.stack(t2.A_d), g['Ad_' + meth], 1e-10)
        close(np.stack(t2.B_d), g['Bd_' + meth], 1e-10)
        close(np.stack(t2.d_d), g['dd_' + meth], 1e-10)
    quiet(tp.pre_discretize, dt)
    # get_jacobians: discrete with the pre-discretised dt, continuous without dt (tpwl.py:251-265)
    for i, x in enumerate(X[:4]):
        A, B, d = tp.get_jacobians(x, dt=dt)
        j = int(g['nearest'][i])
        assert tp.get_ref_point() == j
        np.testing.assert_array_equal(A, tp.A_d[j]); np.testing.assert_array_equal(B, tp.B_d[j])
        np.testing.assert_array_equal(d, tp.d_d[j])
        A, B, d = tp.get_jacobians(x)
        np.testing.assert_array_equal(A, model['A_c'][j])
    xr, zr = tp.rollout(g['roll_x0'], g['roll_u'], dt)
    close(xr, g['roll_x']); close(zr, g['roll_z'])
    from sofacontrol_amd.scp.models.tpwl import TPWLGuSTO
    gm = TPWLGuSTO(tp)
    xc, fc = gm.get_characteristic_vals()
    close(xc, g['x_char'], 1e-13); close(fc, g['f_char'], 1e-12)
    close(tp.get_characteristic_dx(dt), g['dx_char'])
    f = np.stack([gm.get_continuous_dynamics(x, u)[0] for x, u in zip(X, g['roll_u'][:12])])
    close(f, g['fc'])
    close(tp.update_state(X[0], g['roll_u'][0], dt), otpwl.rollout(model, g['Ad_zoh'], g['Bd_zoh'], g['dd_zoh'], X[0], g['roll_u'][:1])[1])


@pytest.mark.parametrize('r,m,P,N,batch', [(30, 4, 64, 50, 5), (30, 8, 64, 50, 3), (5, 4, 9, 10, 4), (36, 4, 100, 20, 2)])
def test_rollout_and_nearest_vs_oracle(r, m, P, N, batch):
    model = otpwl.synthetic_model(r, m, P, seed=r + m)
    model['q'] *= 0.1
    rng = np.random.default_rng(1)
    U, _ = np.linalg.qr(rng.standard_normal((3 * 40, r)))
    from helpers import tip_selector
    tp = product_tpwl(model, U, np.zeros(120), np.zeros(120), tip_selector(7, 40))
    dt = 0.05
    quiet(tp.pre_discretize, dt)
    Ad, Bd, dd = np.stack(tp.A_d), np.stack(tp.B_d), np.stack(tp.d_d)
    X = np.concatenate((0.3 * rng.standard_normal((200, r)), 0.3 * rng.standard_normal((200, r))), axis=1)
    assert np.array_equal(tp.calc_nearest_point(X), otpwl.nearest_points(model, X))
    x0 = 0.01 * rng.standard_normal((batch, 2 * r))
    u = rng.uniform(0, 800, (batch, N, m))
    Xr, Zr = tp.rollout(x0, u, dt)
    for b in range(batch):
        xo = otpwl.rollout(model, Ad, Bd, dd, x0[b], u[b])
        close(Xr[b], xo, 1e-10)
        close(Zr[b], (tp.H @ xo.T).T + tp.z_ref, 1e-10)


def test_weighting_mode_golden(golden):
    """tpwl_method='weighting' (tpwl.py:170-191, 244-250): weights, blended Jacobians, discretised blend and
    the step-wise rollout against the imported reference."""
    g = golden('g3_tpwl')
    model, U, q_ref, v_ref, Hf = golden_problem(4, 3, 7, 20, 10)
    model['w_v'] = 0.5
    tw = product_tpwl(model, U, q_ref, v_ref, Hf, method='weighting', beta=3.0)
    X = g['X']
    W = tw.calc_weighting_factors(X)
    close(W, g['weights'], 1e-13)
    close(tw.calc_weighting_factors(X[3]), g['weights'][3], 1e-15)      # exactly on a point: one-hot
    assert tw.calc_weighting_factors(X[3])[2] == 1.0
    A, B, d, W2 = tw.linearize_batch(X)
    close(A, g['Aw'], 1e-12); close(B, g['Bw'], 1e-12); close(d, g['dw'], 1e-12)
    np.testing.assert_array_equal(W, W2)
    for i in range(4):
        Ad, Bd, dd = tw.get_jacobians(X[i], dt=0.05)
        close(Ad, g['Adw'][i], 1e-11); close(Bd, g['Bdw'][i], 1e-11); close(dd, g['ddw'][i], 1e-11)
    xr, zr = tw.rollout(g['rollw_x'][0], g['rollw_u'], 0.05)
    close(xr, g['rollw_x'], 1e-10); close(zr, g['rollw_z'], 1e-10)
    with pytest.raises(RuntimeError):
        tw.pre_discretize(0.05)
    with pytest.raises(RuntimeError):
        product_tpwl(model, U, q_ref, v_ref, Hf, method='weighting', beta=None)


def test_weighting_gusto_host_loop():
    """GuSTO over a weighting-mode TPWL model goes through the generic host loop around the device QP and
    agrees with the restated loop (oracle) on the same model."""
    from oracle import gusto as ogusto, locp as olocp
    from sofacontrol_amd.scp.models.tpwl import TPWLGuSTO
    from sofacontrol_amd.scp.gusto import GuSTO
    r, m, P, N, dt = 3, 2, 5, 6, 0.05
    model, U, q_ref, v_ref, Hf = golden_problem(r, m, P, 12, 31, q_scale=0.2)
    tw = product_tpwl(model, U, q_ref, v_ref, Hf, method='weighting', beta=2.0)
    gm = TPWLGuSTO(tw)
    H = np.asarray(tw.H)
    Qz = np.diag([0., 0., 0., 100., 100., 0.]); R = 1e-3 * np.eye(m)
    x0 = 0.05 * np.random.default_rng(3).standard_normal(2 * r)
    u_init = np.zeros((N, m))
    x_init, _ = tw.rollout(x0, u_init, dt)
    z = np.tile(H @ x0 + np.array([0, 0, 0, 0.02, -0.01, 0]), (N + 1, 1))
    g = GuSTO(gm, N, dt, Qz, R, x0, u_init, x_init, z=z, verbose=0, max_gusto_iters=4, convg_thresh=1e-3)
    assert not g._fused
    xopt, uopt, zopt, _ = g.get_solution()

    class OModel:
        n_x, n_u = 2 * r, m
    def dyn_d(x, u):
        return otpwl.weighted_jacobians(model, x, 2.0, dt, 'zoh')
    def dyn_c(x, u):
        A, B, d = otpwl.weighted_jacobians(model, x, 2.0)
        return A @ x + B @ u + d, A, B
    xc, fc = gm.get_characteristic_vals()
    xo, uo, _, tr = ogusto.solve_generic(dyn_d, dyn_c, H, N, dt, Qz, R, x0, u_init, x_init, z=z, x_char=xc, f_char=fc,
                                         convg_thresh=1e-3, max_gusto_iters=4)
    close(xopt, xo, 1e-6); close(uopt, uo, 1e-5)


@pytest.mark.parametrize('method', ['fe', 'be', 'bil', 'zoh'])
@pytest.mark.parametrize('n,m,batch', [(8, 3, 5), (60, 4, 7), (72, 8, 3), (10, 8, 2)])
def test_device_discretisation_matches_the_reference_formulas(method, n, m, batch):
    """csrc/discretize.hip (stpwl_discretize) against the restated reference formulas (oracle.tpwl.discretize: tpwl.py:272-297 with
    its two inverses; zoh through scipy.linalg.expm as utils.py:302-335 does): second-order FEM-like models [[-D, -K], [I, 0]] with
    stiff and soft modes, so that zoh needs several squarings and be / bil pivot."""
    import ctypes as C
    from sofacontrol_amd import _lib
    rng = np.random.default_rng(1000 * n + m)
    h = n // 2
    A = np.zeros((batch, n, n)); B = rng.standard_normal((batch, n, m)); d = rng.standard_normal((batch, n))
    for b in range(batch):
        Q, _ = np.linalg.qr(rng.standard_normal((h, h)))
        K = Q @ np.diag(np.logspace(0, 4.5, h)) @ Q.T                     # stiffness spectrum over 4.5 decades
        D = 0.02 * K + 0.5 * np.eye(h)
        A[b, :h, :h], A[b, :h, h:], A[b, h:, :h] = -D, -K, np.eye(h)
    B[:, h:] = 0.0
    dt = 0.05
    Ad = np.empty_like(A); Bd = np.empty_like(B); dd = np.empty_like(d)
    code = {'fe': 0, 'be': 1, 'bil': 2, 'zoh': 3}[method]
    _lib.check(_lib.lib().stpwl_discretize(C.c_int(code), C.c_int(n), C.c_int(m), C.c_int64(batch), _lib.dptr(A), _lib.dptr(B), _lib.dptr(d),
                                           C.c_double(dt), _lib.dptr(Ad), _lib.dptr(Bd), _lib.dptr(dd)), 'stpwl_discretize')
    for b in range(batch):
        Ae, Be, de = otpwl.discretize(A[b], B[b], d[b], dt, method)
        tol = 1e-15 if method == 'fe' else 1e-10            # (be / bil: the reference's inv(A) route loses cond(A) eps ~ 1e-11 itself)
        close(Ad[b], Ae, tol); close(Bd[b], Be, tol); close(dd[b], de, tol)
    if method == 'zoh':
        # a size-independent property: the exponential of twice the step is the square of the exponential
        Ad2 = np.empty_like(A); Bd2 = np.empty_like(B); dd2 = np.empty_like(d)
        _lib.check(_lib.lib().stpwl_discretize(C.c_int(3), C.c_int(n), C.c_int(m), C.c_int64(batch), _lib.dptr(A), _lib.dptr(B), _lib.dptr(d),
                                               C.c_double(2 * dt), _lib.dptr(Ad2), _lib.dptr(Bd2), _lib.dptr(dd2)), 'stpwl_discretize')
        close(Ad2, Ad @ Ad, 1e-11)
        close(Bd2, Ad @ Bd + Bd, 1e-11)


def test_device_discretisation_reports_a_singular_model():
    import ctypes as C
    from sofacontrol_amd import _lib
    A = 2.0 * np.eye(4)[None]                                    # I - dt A = 0 at dt = 0.5
    B = np.ones((1, 4, 2)); d = np.ones((1, 4))
    Ad = np.empty_like(A); Bd = np.empty_like(B); dd = np.empty_like(d)
    rc = _lib.lib().stpwl_discretize(C.c_int(1), C.c_int(4), C.c_int(2), C.c_int64(1), _lib.dptr(A), _lib.dptr(B), _lib.dptr(d),
                                     C.c_double(0.5), _lib.dptr(Ad), _lib.dptr(Bd), _lib.dptr(dd))
    assert rc != 0
    with pytest.raises(Exception, match='singular'):
        _lib.check(rc, 'stpwl_discretize')
    rc = _lib.lib().stpwl_discretize(C.c_int(7), C.c_int(4), C.c_int(2), C.c_int64(1), _lib.dptr(A), _lib.dptr(B), _lib.dptr(d),
                                     C.c_double(0.05), _lib.dptr(Ad), _lib.dptr(Bd), _lib.dptr(dd))
    assert rc != 0


def test_device_discretisation_edge_shapes():
    """One state and one input (a 3 x 3 exponential padded to one tile), an empty batch, and a stiff model whose exponential needs ~17
    squarings (|A| dt ~ 5e5): against scipy / the closed form."""
    import ctypes as C
    from scipy.linalg import expm
    from sofacontrol_amd import _lib
    L = _lib.lib()
    A = np.array([[[-3.0]]]); B = np.array([[[2.0]]]); d = np.array([[0.5]])
    Ad = np.empty_like(A); Bd = np.empty_like(B); dd = np.empty_like(d)
    _lib.check(L.stpwl_discretize(C.c_int(3), C.c_int(1), C.c_int(1), C.c_int64(1), _lib.dptr(A), _lib.dptr(B), _lib.dptr(d), C.c_double(0.1),
                                  _lib.dptr(Ad), _lib.dptr(Bd), _lib.dptr(dd)), 'stpwl_discretize')
    ea = np.exp(-0.3)
    close(Ad, np.array([[[ea]]]), 1e-15); close(Bd, np.array([[[2.0 * (1 - ea) / 3.0]]]), 1e-15); close(dd, np.array([[0.5 * (1 - ea) / 3.0]]), 1e-15)
    assert L.stpwl_discretize(C.c_int(3), C.c_int(1), C.c_int(1), C.c_int64(0), _lib.dptr(A), _lib.dptr(B), _lib.dptr(d), C.c_double(0.1),
                              _lib.dptr(Ad), _lib.dptr(Bd), _lib.dptr(dd)) == 0
    rng = np.random.default_rng(9)
    n, m = 12, 2
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = (Q @ np.diag(-np.logspace(0, 7, n)) @ Q.T)[None]                      # stable, stiff: eigenvalues -1 .. -1e7
    B = rng.standard_normal((1, n, m)); d = rng.standard_normal((1, n))
    Ad = np.empty_like(A); Bd = np.empty_like(B); dd = np.empty_like(d)
    _lib.check(L.stpwl_discretize(C.c_int(3), C.c_int(n), C.c_int(m), C.c_int64(1), _lib.dptr(A), _lib.dptr(B), _lib.dptr(d), C.c_double(0.05),
                                  _lib.dptr(Ad), _lib.dptr(Bd), _lib.dptr(dd)), 'stpwl_discretize')
    M = np.zeros((n + m + 1, n + m + 1)); M[:n, :n] = A[0]; M[:n, n:n + m] = B[0]; M[:n, n + m] = d[0]
    Z = expm(M * 0.05)
    close(Ad[0], Z[:n, :n], 1e-9); close(Bd[0], Z[:n, n:n + m], 1e-9); close(dd[0], Z[:n, n + m], 1e-9)
