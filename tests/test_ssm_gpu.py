"""GPU parity: SSM polynomial model (sofacontrol/SSM/ssm.py) through the C ABI against the golden vectors of the
imported reference (g10_ssm: numpy for jax.numpy, complex-step differentiation for jax.jacobian) and the oracle;
GuSTO over the SSM model (nonlinear observer: per-stage H_d, c_d in the LOCP) against the restated loop."""
import numpy as np
import pytest

from oracle import ssm as ossm, gusto as ogusto

pytestmark = pytest.mark.gpu

CASES = dict(a=(4, 2, 3, 2), b=(6, 4, 3, 3))


def close(a, b, rtol=1e-11):
    np.testing.assert_allclose(a, b, rtol=0, atol=rtol * max(1.0, float(np.abs(b).max())))


def _mat(v):
    a = np.empty((1, 1), dtype=object)
    a[0, 0] = np.asarray(v)
    return a


def product_ssm(model, discrete=False, discr='fe'):
    from sofacontrol_amd.SSM.ssm import SSMDynamics
    sc = lambda v: _mat(np.array([[v]]))
    ro, so = int(model['Er'].sum(axis=1).max()), int(model['Es'].sum(axis=1).max())
    n, m = model['n'], model['m']
    params = dict(state_dim=sc(n), input_dim=sc(m), output_dim=sc(n), SSM_order=sc(so), ROM_order=sc(ro))
    mdl = dict(Ts=sc(0.01), w_coeff=_mat(model['W']), v_coeff=_mat(model['V']), r_coeff=_mat(model['R']),
               B=_mat(model['B']), rd_coeff=_mat(model['Rd']), Bd=_mat(model['Bd']))
    return SSMDynamics(model['z_ref'].copy(), discrete=discrete, discr_method=discr, model=mdl, params=params)


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_golden_g10(golden, tag):
    g = golden('g10_ssm')
    n, m, ro, so = CASES[tag]
    model = ossm.synthetic(n, m, ro, so, seed=60 + n)
    s = product_ssm(model)
    X, U = g[tag + '_X'], g[tag + '_U']
    dt = 0.01
    np.testing.assert_array_equal(s.rom_phi.exponents, model['Er'])
    np.testing.assert_array_equal(s.ssm_phi.exponents, model['Es'])
    close(np.stack([s.rom_phi(*x) for x in X]), g[tag + '_phi_rom'], 1e-14)
    A, B, d = s.get_continuous_jacobians(X, U)
    close(A, g[tag + '_Ac']); close(B, g[tag + '_Bc']); close(d, g[tag + '_dc'])
    A1, B1, d1 = s.get_continuous_jacobians(X[0], U[0])
    np.testing.assert_array_equal(A1, A[0]); np.testing.assert_array_equal(d1, d[0])
    close(s.reduced_dynamics(X, U), g[tag + '_f'])
    H, c = s.get_observer_jacobians(X)
    close(H, g[tag + '_Hobs']); close(c, g[tag + '_cobs'])
    close(np.stack([s.update_observer_state(x) for x in X]), g[tag + '_zobs'])
    close(s.x_to_zfyf(X), g[tag + '_zf'])
    close(s.compute_RO_state(g[tag + '_zf']), g[tag + '_xred'])
    close(s.compute_RO_state(g[tag + '_zf'][1]), g[tag + '_xred'][1])
    for meth in ('fe', 'be', 'bil'):
        sm = product_ssm(model, discr=meth)
        A, B, d = sm.get_jacobians(X, U, dt)
        close(A, g[tag + '_Ad_' + meth], 1e-10); close(B, g[tag + '_Bd_' + meth], 1e-10)
        close(d, g[tag + '_dd_' + meth], 1e-10)
        xr, zr = sm.rollout(X[0], g[tag + '_roll_u'], dt)
        close(xr, g[tag + '_roll_x_' + meth], 1e-10); close(zr, g[tag + '_roll_z_' + meth], 1e-10)
        close(sm.update_state(X[1], U[1], dt), ossm.rollout(model, X[1], U[1:2], dt, meth)[0][1], 1e-10)
    sd = product_ssm(model, discrete=True)
    A, B, d = sd.get_jacobians(X, U, dt)
    close(A, g[tag + '_Ad_map']); close(B, g[tag + '_Bd_map']); close(d, g[tag + '_dd_map'])
    xr, zr = sd.rollout(X[0], g[tag + '_roll_u'], dt)
    close(xr, g[tag + '_roll_x_map']); close(zr, g[tag + '_roll_z_map'])
    from sofacontrol_amd.scp.models.ssm import SSMGuSTO
    gm = SSMGuSTO(s)
    close(np.stack([gm.get_continuous_dynamics(x, u)[0] for x, u in zip(X, U)]), g[tag + '_fc'])
    assert int(g[tag + '_zoh_raises']) == 1
    with pytest.raises(RuntimeError):
        product_ssm(model, discr='zoh').get_jacobians(X[0], U[0], dt)


def test_batched_rollout_vs_oracle():
    n, m = 6, 4
    model = ossm.synthetic(n, m, 3, 3, seed=77)
    s = product_ssm(model, discr='be')
    rng = np.random.default_rng(4)
    x0 = 0.3 * rng.standard_normal((7, n)); u = rng.standard_normal((7, 12, m))
    X, Z = s.rollout(x0, u, 0.01)
    for b in range(7):
        xo, zo = ossm.rollout(model, x0[b], u[b], 0.01, 'be')
        close(X[b], xo, 1e-10); close(Z[b], zo, 1e-10)


@pytest.mark.parametrize('path', ['device', 'host'])
@pytest.mark.parametrize('with_X', [False, True])
def test_gusto_ssm_nonlinear_observer(with_X, path, monkeypatch):
    """GuSTO over SSMGuSTO: per-stage observer linearisation in the QP (locp.py:231-245, 312-329).  `device`: the whole solve in
    one launch of csrc/gusto_ssm.hip (round 6); `host`: the SCP rules on the host around the device QP (SRH_GUSTO_SSM_HOST_LOOP=1,
    what every other TemplateModel runs).  Both against oracle.gusto.solve_generic: equal iteration counts, equal trajectories."""
    if path == 'host':
        monkeypatch.setenv('SRH_GUSTO_SSM_HOST_LOOP', '1')
    from sofacontrol_amd.scp.models.ssm import SSMGuSTO
    from sofacontrol_amd.scp.gusto import GuSTO
    from sofacontrol_amd.utils import HyperRectangle, Polyhedron
    n, m, N, dt = 4, 2, 8, 0.02
    model = ossm.synthetic(n, m, 3, 2, seed=81)
    s = product_ssm(model, discr='fe')
    gm = SSMGuSTO(s)
    rng = np.random.default_rng(9)
    x0 = 0.2 * rng.standard_normal(n)
    u_init = np.zeros((N, m))
    x_init, _ = s.rollout(x0, u_init, dt)
    Qz = np.diag([10., 10., 1., 1.]); R = 1e-2 * np.eye(m)
    z = np.tile(ossm.observe(model, x0) + np.array([0.1, -0.05, 0, 0]), (N + 1, 1))
    U = HyperRectangle([2.0] * m, [-2.0] * m)
    X = None
    if with_X:
        X = Polyhedron(np.array([[1.0, 0, 0, 0], [-1.0, 0, 0, 0]]), np.array([0.5, 0.5]))
    g = GuSTO(gm, N, dt, Qz, R, x0, u_init, x_init, z=z, U=U, X=X, verbose=0, max_gusto_iters=6, convg_thresh=1e-4)
    assert g._ssm == (path == 'device') and g._fused == (path == 'device') and g.nonlinear_observer
    xopt, uopt, zopt, _ = g.get_solution()

    def dyn_d(x, u):
        return ossm.jacobians(model, x, u, dt, 'fe')

    def dyn_c(x, u):
        A, B, d = ossm.continuous_jacobians(model, x, u)
        return A @ x + B @ u + d, A, B
    xo, uo, _, tr = ogusto.solve_generic(dyn_d, dyn_c, np.zeros((n, n)), N, dt, Qz, R, x0, u_init, x_init, z=z,
                                         U=(U.A, U.b), X=None if X is None else (X.A, X.b),
                                         obs_lin=lambda x: ossm.observer_jacobians(model, x),
                                         convg_thresh=1e-4, max_gusto_iters=6)
    assert len(tr) == int(g.iters[0])
    close(xopt, xo, 1e-6); close(uopt, uo, 1e-5)
    np.testing.assert_array_equal(zopt, np.zeros((N + 1, n)))        # gusto.py:483 with H = 0 (ssm.py:69)
    # the (J, delta, omega, rho) trace of every SCP iteration follows the oracle's too
    got = g.trace[0, :len(tr)]
    close(got[:, :3], np.asarray(tr)[:, :3], 1e-6)


@pytest.mark.parametrize('qp', ['dense', 'fused'])
def test_gusto_ssm_three_cost_outputs_follow_the_oracle(qp, monkeypatch):
    """The cost of the reference's SSM hardware driver weighs THREE outputs (examples/hardware/diamond_SSM.py:322-326): no p_o = 2 output
    space for the lean one-wave interior point.  `dense`: the QP in the space of the inputs on one wave (csrc/locp_dense_u.h: N n_u <= 16);
    `fused`: qp::solve (SRH_GUSTO_SSM_NO_DENSE=1).  n_x = 6, n_u = 4, N = 3, U box, up to 4 SCP iterations: iteration counts, (J, delta,
    omega) trace and trajectories of oracle.gusto.solve_generic; with the solver state kept between solves the second solve of the same
    problem returns the same plan."""
    from sofacontrol_amd.scp.models.ssm import SSMGuSTO
    from sofacontrol_amd.scp.gusto import GuSTO
    from sofacontrol_amd.utils import HyperRectangle
    if qp == 'fused':
        monkeypatch.setenv('SRH_GUSTO_SSM_NO_DENSE', '1')
    n, m, N, dt = 6, 4, 3, 0.02
    model = ossm.synthetic(n, m, 3, 2, seed=96)
    s = product_ssm(model, discr='be')
    gm = SSMGuSTO(s)
    rng = np.random.default_rng(5)
    x0 = 0.05 * rng.standard_normal(n)
    u_init = np.zeros((N, m))
    x_init, _ = s.rollout(x0, u_init, dt)
    Qz = np.zeros((n, n)); Qz[0, 0] = Qz[1, 1] = Qz[2, 2] = 100.0
    R = 1e-3 * np.eye(m)
    z = np.tile(ossm.observe(model, x0) + np.array([0.02, -0.01, 0.015, 0, 0, 0]), (N + 1, 1))
    U = HyperRectangle([3.0] * m, [-1.0] * m)
    g = GuSTO(gm, N, dt, Qz, R, x0, u_init, x_init, z=z, U=U, verbose=0, max_gusto_iters=4, convg_thresh=1e-5, keep_solver_state=True)
    assert g._ssm and g.solver_state_kept
    xo, uo, _, tr = ogusto.solve_generic(lambda x, u: ossm.jacobians(model, x, u, dt, 'be'),
                                         lambda x, u: (lambda A, B, d: (A @ x + B @ u + d, A, B))(*ossm.continuous_jacobians(model, x, u)),
                                         np.zeros((n, n)), N, dt, Qz, R, x0, u_init, x_init, z=z, U=(U.A, U.b),
                                         obs_lin=lambda x: ossm.observer_jacobians(model, x), convg_thresh=1e-5, max_gusto_iters=4)
    assert len(tr) == int(g.iters[0])
    close(g.xopt, xo, 1e-6); close(g.uopt, uo, 1e-5)
    close(g.trace[0, :len(tr), :3], np.asarray(tr)[:, :3], 1e-6)
    first = (g.xopt.copy(), g.uopt.copy(), int(g.iters[0]))
    g.solve(x0, u_init, x_init, z, None, None)              # warm: first QP from the previous solve's minimiser and multipliers
    assert int(g.iters[0]) == first[2]
    close(g.xopt, first[0], 1e-7); close(g.uopt, first[1], 1e-6)


def test_gusto_ssm_without_inequality_rows(monkeypatch):
    """No U, no X: the QP of every SCP iteration is an equality-constrained least-squares problem -- the one-wave QP in the space of the
    inputs returns its unit-weight Newton point (no interior-point iteration), the fused path the same; both equal the oracle."""
    from sofacontrol_amd.scp.models.ssm import SSMGuSTO
    from sofacontrol_amd.scp.gusto import GuSTO
    n, m, N, dt = 6, 4, 3, 0.02
    model = ossm.synthetic(n, m, 3, 2, seed=96)
    rng = np.random.default_rng(8)
    x0 = 0.05 * rng.standard_normal(n)
    u_init = np.zeros((N, m))
    Qz = np.zeros((n, n)); Qz[0, 0] = Qz[1, 1] = Qz[2, 2] = 100.0
    R = 1e-2 * np.eye(m)
    res = {}
    for qp in ('dense', 'fused'):
        if qp == 'fused':
            monkeypatch.setenv('SRH_GUSTO_SSM_NO_DENSE', '1')
        s = product_ssm(model, discr='be')
        gm = SSMGuSTO(s)
        x_init, _ = s.rollout(x0, u_init, dt)
        z = np.tile(ossm.observe(model, x0) + np.array([0.01, -0.01, 0.005, 0, 0, 0]), (N + 1, 1))
        g = GuSTO(gm, N, dt, Qz, R, x0, u_init, x_init, z=z, verbose=0, max_gusto_iters=3, convg_thresh=1e-6)
        assert g._ssm
        res[qp] = (g.xopt.copy(), g.uopt.copy(), int(g.iters[0]))
    monkeypatch.delenv('SRH_GUSTO_SSM_NO_DENSE', raising=False)
    xo, uo, _, tr = ogusto.solve_generic(lambda x, u: ossm.jacobians(model, x, u, dt, 'be'),
                                         lambda x, u: (lambda A, B, d: (A @ x + B @ u + d, A, B))(*ossm.continuous_jacobians(model, x, u)),
                                         np.zeros((n, n)), N, dt, Qz, R, x0, u_init, x_init, z=z,
                                         obs_lin=lambda x: ossm.observer_jacobians(model, x), convg_thresh=1e-6, max_gusto_iters=3)
    for qp in ('dense', 'fused'):
        assert res[qp][2] == len(tr), (qp, res[qp][2], len(tr))
        close(res[qp][0], xo, 1e-6); close(res[qp][1], uo, 1e-5)


def test_gusto_ssm_real_time_iteration_device_equals_host_loop(monkeypatch):
    """The reference's hardware loop (examples/hardware/diamond_SSM.py:353-361: n_x = 6, n_u = 4, N = 3, dt = 0.02,
    max_gusto_iters = 0 -- one QP per call) on the device path and on the host loop: same plans over a sequence of receding-horizon
    calls (the constructor's solve runs the default cap of 500, gusto.py:142-147), batched device solves equal the single ones."""
    from sofacontrol_amd.scp.models.ssm import SSMGuSTO
    from sofacontrol_amd.scp.gusto import GuSTO
    from sofacontrol_amd.utils import HyperRectangle
    n, m, N, dt = 6, 4, 3, 0.02
    model = ossm.synthetic(n, m, 3, 2, seed=96)
    rng = np.random.default_rng(12)
    Qz = np.zeros((n, n)); Qz[0, 0] = Qz[1, 1] = 100.0
    R = 0.003 * np.eye(m)
    U = HyperRectangle([1500.0] * m, [0.0] * m)
    x0s = 0.05 * rng.standard_normal((4, n))
    z = np.tile(np.array([0.02, -0.01, 0, 0, 0, 0.0]), (N + 1, 1))
    res = {}
    for path in ('device', 'host'):
        if path == 'host':
            monkeypatch.setenv('SRH_GUSTO_SSM_HOST_LOOP', '1')
        else:
            monkeypatch.delenv('SRH_GUSTO_SSM_HOST_LOOP', raising=False)
        s = product_ssm(model, discr='be')
        gm = SSMGuSTO(s)
        u0 = np.zeros((N, m))
        xi, _ = s.rollout(x0s[0], u0, dt)
        g = GuSTO(gm, N, dt, Qz, R, x0s[0], u0, xi, z=z, U=U, verbose=0, max_gusto_iters=0, convg_thresh=1e-3)
        assert g._ssm == (path == 'device')
        out = [(g.xopt.copy(), g.uopt.copy(), int(g.iters[0]))]
        for b in range(1, 4):
            xi, _ = s.rollout(x0s[b], out[-1][1], dt)
            g.solve(x0s[b], out[-1][1], xi, z, None, None)
            assert int(g.iters[0]) == 1
            out.append((g.xopt.copy(), g.uopt.copy(), 1))
        res[path] = out
    for a, b in zip(res['device'], res['host']):
        assert a[2] == b[2]
        close(a[0], b[0], 1e-7); close(a[1], b[1], 1e-6)
    # a batch of four rollouts in one launch equals four single solves
    monkeypatch.delenv('SRH_GUSTO_SSM_HOST_LOOP', raising=False)
    s = product_ssm(model, discr='be')
    gm = SSMGuSTO(s)
    u0 = np.zeros((4, N, m))
    xi = np.stack([s.rollout(x0s[b], u0[b], dt)[0] for b in range(4)])
    zb = np.tile(z, (4, 1, 1))
    gb = GuSTO(gm, N, dt, Qz, R, x0s, u0, xi, z=zb, U=U, verbose=0, max_gusto_iters=3, convg_thresh=1e-3, batch=4, first_solve_cap=3)
    g1 = GuSTO(gm, N, dt, Qz, R, x0s[0], u0[0], xi[0], z=z, U=U, verbose=0, max_gusto_iters=3, convg_thresh=1e-3, first_solve_cap=3)
    for b in range(4):
        g1.solve(x0s[b], u0[b], xi[b], z, None, None)
        assert int(g1.iters[0]) == int(gb.iters[b])
        close(gb.xopt[b], g1.xopt, 1e-12); close(gb.uopt[b], g1.uopt, 1e-12)


ILQR_SSM_CASES = dict(h0=('be', 20, False), hw=('be', 40, True), fe=('fe', 25, True))


@pytest.mark.parametrize('tag', sorted(ILQR_SSM_CASES))
def test_ilqr_ssm_golden(golden, tag):
    """iLQR over the SSM model in one kernel (silqr_solve_ssm) against the imported reference (g11)."""
    from sofacontrol_amd.lqr.ilqr import iLQR
    from sofacontrol_amd.utils import QuadraticCost
    g = golden('g11_ilqr_ssm')
    meth, N, useH = ILQR_SSM_CASES[tag]
    model = ossm.synthetic(6, 4, 3, 3, seed=90)
    s = product_ssm(model, discr=meth)
    if useH:
        s.H = model['W'][:, :6].copy()
    Qz = g[tag + '_Qz']
    il = iLQR(0.01, s, QuadraticCost(Q=Qz, R=0.05 * np.eye(4), Qf=5 * Qz), N)
    il.set_target(g[tag + '_z_target'])
    x, u, K = il.ilqr_computation(g[tag + '_x0'], g[tag + '_uw'])
    assert int(il.iters[0]) == int(g[tag + '_iters'])
    close(x, g[tag + '_x'], 1e-9); close(u, g[tag + '_u'], 1e-8); close(K, g[tag + '_K'], 1e-7)


def test_ilqr_ssm_c3_shape_batched():
    """BASELINE config C3 exactly as bench.py times it (workloads.ssm_c3: SSM r = 10, n_u = 8, horizon 100, dt = 0.05,
    backward Euler -- the in-kernel LDS Gauss-Jordan branch), the first problems of the bench's batch, vs the oracle loop:
    identical iteration counts, costs to 1e-5 relative.  Trajectories: this configuration is ill conditioned IN THE REFERENCE
    ALGORITHM -- a relative perturbation of 1e-14 of x0 moves the oracle's own result by 1e-6 .. 1e-5 in x, up to 7e-5 in u and
    1e-3 in the gains (the un-symmetrised Riccati recursion of ilqr.py:219-300 over 100 backward-Euler stages amplifies rounding
    by ~1e10; at dt = 0.01 / forward Euler, `test_ilqr_ssm_fe_small_step_exact`, device and oracle agree to 1e-15).  The
    tolerance is therefore MEASURED here: ten times what that perturbation does to the oracle, and never above the 1e-4 of
    the north star for the state trajectory."""
    import workloads as wl
    from oracle import lqr as olqr
    from sofacontrol_amd.lqr.ilqr import iLQR
    from sofacontrol_amd.utils import QuadraticCost
    c3 = wl.ssm_c3(256)
    Bn = 3
    n, m, N, dt, discr = c3['n'], c3['m'], c3['N'], c3['dt'], c3['discr']
    assert (dt, discr) == (0.05, 'be')
    model = ossm.synthetic(n, m, 3, 2, seed=95)
    np.testing.assert_array_equal(model['R'], c3['model']['R'])
    s = product_ssm(model, discr=discr)
    s.H = model['W'][:, :n].copy()
    Qz, R, x0, zt = c3['Qz'], c3['R'], c3['x0'][:Bn], c3['zt'][:Bn]
    il = iLQR(dt, s, QuadraticCost(Q=Qz, R=R, Qf=c3['Qf']), N)
    il.set_target(zt)
    x, u, K = il.ilqr_computation(x0)

    def oracle(xs, b):
        o = olqr.ILQRGeneric(lambda xx, uu: ossm.jacobians(model, xx, uu, dt, discr),
                             lambda xx: ossm.observe(model, xx) + model['z_ref'], s.H, n, m, Qz, R, c3['Qf'], N)
        xo, uo, Ko = o.solve(xs, zt[b])
        return xo, uo, o
    for b in range(Bn):
        xo, uo, o = oracle(x0[b], b)
        xp, up, _ = oracle(x0[b] * (1.0 + 1e-14), b)
        sx, su = np.abs(xo - xp).max(), np.abs(uo - up).max()
        assert int(il.iters[b]) == len(o.trace) - 1
        assert abs(float(il.cost[b]) - o.trace[-1][1]) <= 1e-5 * abs(o.trace[-1][1])
        ex, eu = np.abs(x[b] - xo).max(), np.abs(u[b] - uo).max()
        assert ex <= max(1e-8, 10.0 * sx) and ex <= 1e-4 * np.abs(xo).max(), (b, ex, sx)
        assert eu <= max(1e-7, 10.0 * su), (b, eu, su)


def test_ilqr_ssm_fe_small_step_exact():
    """The same C3 problems at dt = 0.01 with forward Euler (a well conditioned recursion): device and oracle agree to
    rounding -- what separates them at the bench's dt = 0.05 / backward Euler is conditioning, not the kernel."""
    import workloads as wl
    from oracle import lqr as olqr
    from sofacontrol_amd.lqr.ilqr import iLQR
    from sofacontrol_amd.utils import QuadraticCost
    c3 = wl.ssm_c3(256)
    n, m, N, dt = c3['n'], c3['m'], c3['N'], 0.01
    model = ossm.synthetic(n, m, 3, 2, seed=95)
    for discr in ('fe', 'be'):
        s = product_ssm(model, discr=discr)
        s.H = model['W'][:, :n].copy()
        il = iLQR(dt, s, QuadraticCost(Q=c3['Qz'], R=c3['R'], Qf=c3['Qf']), N)
        il.set_target(c3['zt'][:2])
        x, u, K = il.ilqr_computation(c3['x0'][:2])
        for b in range(2):
            o = olqr.ILQRGeneric(lambda xx, uu: ossm.jacobians(model, xx, uu, dt, discr),
                                 lambda xx: ossm.observe(model, xx) + model['z_ref'], s.H, n, m, c3['Qz'], c3['R'], c3['Qf'], N)
            xo, uo, Ko = o.solve(c3['x0'][b], c3['zt'][b])
            assert int(il.iters[b]) == len(o.trace) - 1
            close(x[b], xo, 1e-10); close(u[b], uo, 1e-9)


@pytest.mark.parametrize('tag', ['discrete', 'be', 'fe'])
def test_shipped_ssm_model_open_loop_rollout(golden, tag):
    """The reference's own SSM module test (examples/hardware/diamond_SSM.py:21-80) on its shipped Diamond model
    (SSM_model.mat: n_x = 6, n_u = 4, cubic) and recorded inputs / outputs (checkModel/u_big.csv, z_big.csv; golden g17
    holds the data files and what the imported reference computes): 1002-step open-loop rollout on the device, the
    trajectories and the mean squared error against the measured tip trajectory."""
    from sofacontrol_amd.SSM.ssm import SSMDynamics
    g = golden('g17_ssm_hardware')
    mdl = {k[len('model_'):]: _mat(g[k]) for k in g.files if k.startswith('model_')}
    prm = {k[len('params_'):]: _mat(g[k]) for k in g.files if k.startswith('params_')}
    kw = dict(discrete=dict(discrete=True, discr_method='be'), be=dict(discrete=False, discr_method='be'),
              fe=dict(discrete=False, discr_method='fe'))[tag]
    s = SSMDynamics(g['z_eq'].copy(), model=mdl, params=prm, **kw)
    p, z = s.rollout(np.zeros(6), g['u_interp'], float(g['dt']))
    close(p, g[tag + '_p'], 1e-8); close(z, g[tag + '_z'], 1e-8)
    err = g['z_true_qv'] - z[:-1]
    mse = np.linalg.norm(np.linalg.norm(err, axis=1)) ** 2 / err.shape[0]
    assert mse == pytest.approx(float(g[tag + '_mse']), rel=1e-8)
