"""CPU: the oracle (numpy restatement) against golden vectors produced by the imported reference
(tests/golden/make_golden.py).  Tolerances: 1e-12 relative for direct algebra, looser where the
reference iterates (stated per test)."""
import numpy as np
import pytest

from oracle import pod as opod, tpwl as otpwl, lqr as olqr, locp as olocp, gusto as ogusto


def close(a, b, rtol=1e-12, atol=1e-12):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * max(1.0, float(np.abs(b).max())))


# ---------------------------------------------------------------- G1: POD
def test_pod_project_lift_reduce(golden):
    g = golden('g1_pod')
    U, q_ref, v_ref = g['U'], g['q_ref'], g['v_ref']
    close(opod.make_V(U), g['V'])
    close(opod.project(U, q_ref, g['Xq']), g['proj_q'])
    close(opod.project(U, v_ref, g['Xv']), g['proj_v'])
    Xx = opod.qv2x(g['Xq'], g['Xv'])
    close(opod.project_x(U, q_ref, v_ref, Xx), g['proj_x'])
    pr = g['proj_x']
    close(opod.lift(U, q_ref, pr[:, 8:]), g['lift_q'])
    close(opod.lift(U, v_ref, pr[:, :8]), g['lift_v'])
    close(opod.lift_x(U, q_ref, v_ref, pr), g['lift_x'])
    close(opod.reduce_matrix(U, g['M']), g['UMU'])
    close(opod.reduce_matrix(U, g['M'], left=True), g['UM'])
    close(opod.reduce_matrix(U, g['M'], right=True), g['MU'])
    close(opod.reduce_matrix(U, g['Mc_dense']), g['UMcU'])
    close(opod.reduce_matrix(U, g['Hm'], left=True), g['UH'])


def test_pod_svd_truncation_and_gramian_route(golden):
    g = golden('g1_pod')
    S = g['pod_S']
    for tol in (1e-2, 1e-4, 1e-7):
        U_full, Uk, k, Sig = opod.compute_pod(S, tol)
        assert k == int(g['pod_k_%g' % tol])
        # method of snapshots (the GPU route) gives the same spectrum / modes up to sign
        U2, k2, Sig2 = opod.pod_from_gramian(S.T.copy(), tol)
        assert k2 == k
        # sigma_i from eig(G) carries a relative error ~ eps*(sigma_0/sigma_i)^2
        close(Sig2[:6], g['pod_Sigma'][:6], rtol=1e-9)
        close(Sig2[:8], g['pod_Sigma'][:8], rtol=1e-5)
        close(np.abs(U2), np.abs(Uk), rtol=1e-6, atol=1e-8)
    close(Sig, g['pod_Sigma'])
    close(np.abs(U_full[:, :8]), g['pod_Ufull_abs'], rtol=1e-8, atol=1e-10)
    # shipped Diamond model: tolerance 5e-5 keeps 36 modes (examples/diamond/pod_model.pkl)
    assert opod.energy_truncation(g['shipped_Sigma'], float(g['shipped_tol'])) == int(g['shipped_k']) == 36


def test_pod_snapshot_types(golden):
    g = golden('g1_pod')
    q_ref = g['q_ref']
    data = dict(q=[q_ref + i * np.ones(300) for i in range(4)], v=[i * np.ones(300) for i in range(4)])
    data['v+'] = [2.0 * i * np.ones(300) for i in range(4)]
    for t in 'qva':
        close(opod.get_snapshots(data, t), g['snap_' + t])


# ---------------------------------------------------------------- G3: TPWL
def _g3_model():
    m = otpwl.synthetic_model(4, 3, 7, seed=10)
    m['w_v'] = 0.5
    return m


def test_tpwl_nearest_weights_jacobians(golden):
    g = golden('g3_tpwl')
    model = _g3_model()
    X = g['X']
    assert np.array_equal(otpwl.nearest_points(model, X), g['nearest'])
    assert otpwl.nearest_point(model, X[3]) == 2
    W = np.stack([otpwl.weighting_factors(model, x, 3.0) for x in X])
    close(W, g['weights'])
    close(np.einsum('bi,ijk->bjk', W, model['A_c']), g['Aw'])
    close(np.einsum('bi,ijk->bjk', W, model['B_c']), g['Bw'])
    close(np.einsum('bi,ij->bj', W, model['d_c']), g['dw'])
    for i in range(4):
        A, B, d = otpwl.weighted_jacobians(model, X[i], 3.0, 0.05, 'zoh')
        close(A, g['Adw'][i]); close(B, g['Bdw'][i]); close(d, g['ddw'][i])
    close(otpwl.rollout_weighted(model, g['rollw_x'][0], g['rollw_u'], 3.0, 0.05), g['rollw_x'])


@pytest.mark.parametrize('meth', ['fe', 'be', 'bil', 'zoh'])
def test_tpwl_discretize(golden, meth):
    g = golden('g3_tpwl')
    Ad, Bd, dd = otpwl.pre_discretize(_g3_model(), 0.05, meth)
    close(Ad, g['Ad_' + meth], rtol=1e-10)
    close(Bd, g['Bd_' + meth], rtol=1e-10)
    close(dd, g['dd_' + meth], rtol=1e-10)


def test_tpwl_rollout_and_characteristics(golden):
    g = golden('g3_tpwl')
    model = _g3_model()
    Ad, Bd, dd = g['Ad_zoh'], g['Bd_zoh'], g['dd_zoh']
    x = otpwl.rollout(model, Ad, Bd, dd, g['roll_x0'], g['roll_u'])
    close(x, g['roll_x'], rtol=1e-11)
    close((g['H'] @ x.T).T + g['z_ref'], g['roll_z'], rtol=1e-11)
    xc, fc = otpwl.characteristic_vals(model)
    close(xc, g['x_char'])
    close(fc, g['f_char'])
    close(otpwl.characteristic_dx(model, Ad, Bd, dd), g['dx_char'])
    f = np.stack([otpwl.continuous_dynamics(model, x_, u_)[0] for x_, u_ in zip(g['X'], g['roll_u'][:12])])
    close(f, g['fc'])
    # H = Hf V (tpwl.py:86-89)
    close(g['Hf'] @ opod.make_V(g['U']), g['H'])
    close(g['Hf'] @ opod.qv2x(g['q_ref'], g['v_ref']), g['z_ref'])


# ---------------------------------------------------------------- G4: Riccati / iLQR
def _g4_setup(golden):
    g = golden('g4_riccati')
    model = otpwl.synthetic_model(5, 4, 9, seed=20)
    Ad, Bd, dd = otpwl.pre_discretize(model, 0.05, 'zoh')
    return g, model, Ad, Bd, dd


def test_dare_and_fixed_point_riccati(golden):
    g, model, Ad, Bd, dd = _g4_setup(golden)
    L, P, _ = olqr.solve_riccati(Ad[2], Bd[2], g['Q'], g['R'])
    close(L, g['sr_L'], rtol=1e-9)
    close(P, g['sr_P'], rtol=1e-9)
    K, P = olqr.dare(Ad[2], Bd[2], g['Q'], g['R'])
    close(K, g['dare_K'], rtol=1e-9)
    close(P, g['dare_P'], rtol=1e-9)


def test_tvlqr_recursion(golden):
    g, model, Ad, Bd, dd = _g4_setup(golden)
    # reference interpolates the nominal at t_i = i*dt: these are the target samples themselves
    xbar = g['tt_xbar']
    close(xbar, g['tt_x'][:-1], rtol=1e-12)
    idx = otpwl.nearest_points(model, xbar)
    K, P = olqr.tvlqr(Ad[idx], Bd[idx], g['Q'], g['R'])
    close(K, g['tt_K'], rtol=1e-9)
    close(P, g['tt_P'], rtol=1e-9)


@pytest.mark.parametrize('tag,N', [('c1', 10), ('n30', 30)])
def test_ilqr_backward_and_full_solve(golden, tag, N):
    g, model, Ad, Bd, dd = _g4_setup(golden)
    il = olqr.ILQR(model, Ad, Bd, dd, g['H'], g['z_ref'], g['Qz'], g['R'], g['Qf'], N)
    il.z_target = g[tag + '_z_target']
    il.rho, il.drho = 0., 0.
    xp = np.zeros((N + 1, 10)); xp[0] = g[tag + '_x0']
    x, u, c, A, B, d = il.forward_pass(xp, g[tag + '_uw'])
    close(x, g[tag + '_fp_x'], rtol=1e-11)
    close(c, g[tag + '_fp_cost'], rtol=1e-11)
    K, k, Qu, Quu = il.dlqr_recursion(x, u, A, B, d)
    close(K, g[tag + '_K'], rtol=1e-8)
    close(k, g[tag + '_k'], rtol=1e-8)
    close(Qu, g[tag + '_Qu'], rtol=1e-8)
    close(Quu, g[tag + '_Quu'], rtol=1e-8)
    assert il.rho == float(g[tag + '_rho_after'])
    xs, us, Ks = il.solve(g[tag + '_x0'], g[tag + '_z_target'], g[tag + '_uw'])
    assert len(il.trace) - 1 == int(g[tag + '_iters'])
    close(xs, g[tag + '_sol_x'], rtol=1e-7)
    close(us, g[tag + '_sol_u'], rtol=1e-7)
    close(Ks, g[tag + '_sol_K'], rtol=1e-7)
    xs, us, Ks = il.solve(g[tag + '_x0'], g[tag + '_z_target'])
    assert len(il.trace) - 1 == int(g[tag + '_iters0'])
    close(xs, g[tag + '_sol0_x'], rtol=1e-7)
    close(us, g[tag + '_sol0_u'], rtol=1e-7)


G20_CASES = ['reference', 'include_input_var_constraint', 'do_linesearch', 'regularize', 'state_regularization', 'all_off']


def _g20_flags(case):
    flags = ('include_input_var_constraint', 'do_linesearch', 'regularize', 'state_regularization')
    return {f: not (case == f or case == 'all_off') for f in flags}


@pytest.mark.parametrize('case', G20_CASES)
def test_ilqr_config_switches(golden, case):
    """g20: the imported reference iLQR with each switch of lqr/config.py:6-9, 31 turned off (and all of them): the oracle
    statement follows -- same iteration counts, trajectories and gains."""
    g4, model, Ad, Bd, dd = _g4_setup(golden)
    g = golden('g20_ilqr_switches')
    N = int(g['N'])
    for warm in (True, False):
        il = olqr.ILQR(model, Ad, Bd, dd, g4['H'], g4['z_ref'], g['Qz'], g['R'], g['Qf'], N)
        for f, v in _g20_flags(case).items():
            setattr(il.p, f, v)
        il.u_last = g['u_last']
        xs, us, Ks = il.solve(g['x0'], g['z_target'], g['uw'] if warm else None)
        key = case + ('_warm' if warm else '_cold')
        assert len(il.trace) - 1 == int(g[key + '_iters']), (key, len(il.trace) - 1, int(g[key + '_iters']))
        close(xs, g[key + '_x'], rtol=1e-7)
        close(us, g[key + '_u'], rtol=1e-7)
        close(Ks, g[key + '_K'], rtol=1e-6)


# ---------------------------------------------------------------- G6: GuSTO
def _g6_setup(golden):
    g = golden('g6_gusto')
    model = otpwl.synthetic_model(4, 3, 7, seed=30)
    model['q'] = model['q'] * 0.05
    Ad, Bd, dd = otpwl.pre_discretize(model, 0.05, 'zoh')
    return g, model, Ad, Bd, dd


def test_gusto_helpers(golden):
    g, model, Ad, Bd, dd = _g6_setup(golden)
    xs, fs = 1. / np.abs(g['x_char']), 1. / np.abs(g['f_char'])
    x, u, xk, uk = g['h_x'], g['h_u'], g['h_xk'], g['h_uk']
    tr = [ogusto.is_in_trust_region(x, xk, xs, d, 0.01)[0] for d in (1e-3, 1e-1, 10.)]
    close(tr, g['h_tr'])
    close(ogusto.is_converged(x, xk, xs, 12, 1e-3)[0], g['h_conv'])
    close(ogusto.compute_accuracy(model, x, u, xk, uk, 3.7, 0.05, fs), g['h_rho'], rtol=1e-10)
    close(ogusto.state_violation((g['Xp_A'], g['Xp_b']), 10 * x), g['h_viol'])
    xc, fc = otpwl.characteristic_vals(model)
    close(xc, g['x_char']); close(fc, g['f_char'])


@pytest.mark.parametrize('tag', ['box', 'boxX', 'free'])
def test_gusto_outer_loop_matches_reference_loop(golden, tag):
    """The reference GuSTO class (imported, with this oracle's exact QP injected for cvxpy's LOCP)
    against the restated loop: same iterates, same (J, delta, omega) sequence."""
    g, model, Ad, Bd, dd = _g6_setup(golden)
    N, dt = 12, 0.05
    cons = {}
    if tag in ('box', 'boxX'):
        cons['U'] = (g['U_A'], g['U_b'])
    if tag == 'boxX':
        cons['X'] = (g['Xp_A'], g['Xp_b'])
    from scipy.interpolate import interp1d
    zi = interp1d(g['t'], g['zt'], axis=0, bounds_error=False, fill_value=(g['zt'][0], g['zt'][-1]))
    x0 = np.zeros(8)
    u_init = np.zeros((N, 3))
    x_init = otpwl.rollout(model, Ad, Bd, dd, x0, u_init)
    z = zi(dt * np.arange(N + 1))
    xo, uo, zo, tr = ogusto.solve(model, Ad, Bd, dd, g['H'], N, dt, g['Qz'], g['R'], x0, u_init, x_init,
                                  z=z, x_char=g['x_char'], f_char=g['f_char'], convg_thresh=1e-3, **cons)
    ref_tr = g[tag + '_trace']
    assert len(tr) == ref_tr.shape[0]
    close(np.array([t[:3] for t in tr]), ref_tr, rtol=1e-6)
    close(xo, g[tag + '_xopt'], rtol=1e-6, atol=1e-8)
    close(uo, g[tag + '_uopt'], rtol=1e-6, atol=1e-8)
    close(zo, g[tag + '_zopt'], rtol=1e-6, atol=1e-8)
    # warm-started re-solve (shifted previous solution, scp/ros.py:109-114)
    x2, u2, z2, tr2 = ogusto.solve(model, Ad, Bd, dd, g['H'], N, dt, g['Qz'], g['R'], g[tag + '_x0b'],
                                   g[tag + '_uinit'], g[tag + '_xinit'], z=g[tag + '_zb'],
                                   x_char=g['x_char'], f_char=g['f_char'], convg_thresh=1e-3, **cons)
    assert len(tr2) == g[tag + '_trace2'].shape[0]
    close(x2, g[tag + '_xopt2'], rtol=1e-5, atol=1e-7)
    close(u2, g[tag + '_uopt2'], rtol=1e-5, atol=1e-7)
    close(zi(0.37 + dt * np.arange(N + 1)), g['get_target_z'])


# ---------------------------------------------------------------- G9: EKF observer
def test_ekf_oracle(golden):
    from oracle import observer as oobs
    from helpers import golden_problem, meas_selector
    g = golden('g9_ekf')
    model, U, q_ref, v_ref, Hf = golden_problem(4, 3, 7, 20, 50, q_scale=0.3)
    Cf = meas_selector([3, 9], 20)
    np.testing.assert_array_equal(Cf.toarray(), g['Cf'])
    V = np.kron(np.eye(2), U)
    x_ref = np.concatenate((v_ref, q_ref))
    C = Cf @ V
    y_ref = Cf @ x_ref
    Ad, Bd, dd = otpwl.pre_discretize(model, 0.02, 'zoh')
    x, S = g['x_init'], g['Sigma0']
    close(x, np.zeros(8))
    for k in range(8):
        x, S = oobs.predict(model, Ad, Bd, dd, x, S, g['u'][k], g['W'])
        x, S = oobs.update(C, y_ref, x, S, g['y'][k], g['V'])
        close(x, g['x'][k], 1e-10); close(S, g['Sigma'][k], 1e-10)
    x, S = oobs.predict(model, Ad, Bd, dd, x, S, g['u'][0], g['W'])
    close(x, g['x_pred'], 1e-10); close(S, g['Sigma_pred'], 1e-10)
    x, S = oobs.update(C, y_ref, x, S, g['y'][1], g['V'])
    close(x, g['x_upd'], 1e-10); close(S, g['Sigma_upd'], 1e-10)


# ---------------------------------------------------------------- G10: SSM polynomial model
@pytest.mark.parametrize('tag,shape', [('a', (4, 2, 3, 2)), ('b', (6, 4, 3, 3))])
def test_ssm_oracle(golden, tag, shape):
    from oracle import ssm as ossm
    g = golden('g10_ssm')
    n, m, ro, so = shape
    model = ossm.synthetic(n, m, ro, so, seed=60 + n)
    X, U = g[tag + '_X'], g[tag + '_U']
    assert model['Er'].shape[0] == g[tag + '_phi_rom'].shape[1]
    close(np.stack([ossm.phi(model['Er'], x) for x in X]), g[tag + '_phi_rom'], 1e-14)
    close(np.stack([ossm.phi(model['Es'], x) for x in X]), g[tag + '_phi_ssm'], 1e-14)
    A, B, d = zip(*[ossm.continuous_jacobians(model, x, u) for x, u in zip(X, U)])
    close(np.stack(A), g[tag + '_Ac']); close(np.stack(B), g[tag + '_Bc']); close(np.stack(d), g[tag + '_dc'])
    close(np.stack([ossm.dynamics(model, x, u) for x, u in zip(X, U)]), g[tag + '_f'])
    for meth in ('fe', 'be', 'bil'):
        A, B, d = zip(*[ossm.jacobians(model, x, u, 0.01, meth) for x, u in zip(X, U)])
        close(np.stack(A), g[tag + '_Ad_' + meth]); close(np.stack(B), g[tag + '_Bd_' + meth])
        close(np.stack(d), g[tag + '_dd_' + meth])
        xr, zr = ossm.rollout(model, X[0], g[tag + '_roll_u'], 0.01, meth)
        close(xr, g[tag + '_roll_x_' + meth]); close(zr, g[tag + '_roll_z_' + meth])
    A, B, d = zip(*[ossm.jacobians(model, x, u, 0.01, discrete=True) for x, u in zip(X, U)])
    close(np.stack(A), g[tag + '_Ad_map']); close(np.stack(d), g[tag + '_dd_map'])
    xr, zr = ossm.rollout(model, X[0], g[tag + '_roll_u'], 0.01, discrete=True)
    close(xr, g[tag + '_roll_x_map']); close(zr, g[tag + '_roll_z_map'])
    H, c = zip(*[ossm.observer_jacobians(model, x) for x in X])
    close(np.stack(H), g[tag + '_Hobs']); close(np.stack(c), g[tag + '_cobs'])
    close(np.stack([ossm.reduce(model, z) for z in g[tag + '_zf']]), g[tag + '_xred'])
    with pytest.raises(RuntimeError):
        ossm.discretize(np.eye(2), np.eye(2), np.ones(2), 0.1, 'zoh')


# ---------------------------------------------------------------- G11: iLQR over the SSM model
ILQR_SSM_CASES = dict(h0=('be', 20, False), hw=('be', 40, True), fe=('fe', 25, True))


@pytest.mark.parametrize('tag', sorted(ILQR_SSM_CASES))
def test_ilqr_ssm_oracle(golden, tag):
    from oracle import ssm as ossm, lqr as olqr
    g = golden('g11_ilqr_ssm')
    meth, N, useH = ILQR_SSM_CASES[tag]
    model = ossm.synthetic(6, 4, 3, 3, seed=90)
    H = model['W'][:, :6] if useH else np.zeros((6, 6))
    Qz = g[tag + '_Qz']
    il = olqr.ILQRGeneric(lambda x, u: ossm.jacobians(model, x, u, 0.01, meth),
                          lambda x: ossm.observe(model, x) + model['z_ref'], H, 6, 4, Qz, 0.05 * np.eye(4), 5 * Qz, N)
    x, u, K = il.solve(g[tag + '_x0'], g[tag + '_z_target'], g[tag + '_uw'])
    assert len(il.trace) - 1 == int(g[tag + '_iters'])
    close(x, g[tag + '_x'], 1e-10); close(u, g[tag + '_u'], 1e-9); close(K, g[tag + '_K'], 1e-8)


# ---------------------------------------------------------------------------------------------------------------
# g14: oracle.locp.build_qp against the reference's OWN statement of the QP -- sofacontrol/scp/locp.py executed
# through the evaluating cvxpy stand-in (tests/golden/_cvxpy_eval.py) in the build container: J(w) of
# locp.py:218-263 and the residual of every constraint of locp.py:265-342 at ten seeded points and at the optimum.
import qp_cases  # noqa: E402


@pytest.mark.parametrize('name', sorted(qp_cases.G14_CASES))
def test_locp_statement_matches_reference_locp_py(golden, name):
    g = golden('g14_locp')
    case = qp_cases.g14_case(name)
    pts = qp_cases.g14_points(name, case)
    ws = [qp_cases.g14_pack(case, *p) for p in pts] + [g[name + '_wopt']]
    Jr, Rr = g[name + '_J'], g[name + '_res']
    assert Jr.shape[0] == len(ws) == Rr.shape[0]
    for i, w in enumerate(ws):
        J, res, qp = qp_cases.g14_oracle_values(case, w)
        assert res.shape == Rr[i].shape, (res.shape, Rr[i].shape)          # same rows in the same order
        np.testing.assert_allclose(J, Jr[i], rtol=1e-12, atol=1e-12 * max(1.0, abs(Jr[i])))
        np.testing.assert_allclose(res, Rr[i], rtol=0, atol=1e-12 * max(1.0, np.abs(Rr[i]).max()))
    # the stored optimum is a KKT point of the oracle QP, and feasible in the REFERENCE's own constraint residuals
    from oracle import locp as olocp
    w = g[name + '_wopt']
    n_eq_dyn = qp.N * qp.n
    r = Rr[-1]
    assert np.abs(r[:n_eq_dyn]).max() <= 1e-8 and np.abs(r[-qp.n:]).max() <= 1e-8      # dynamics and x_0 = x0
    assert r[n_eq_dyn:-qp.n].max(initial=0.0) <= 1e-7                                     # every inequality


@pytest.mark.parametrize('name', sorted(qp_cases.NULLSPACE_CASES))
def test_locp_input_nullspace_statement_and_optimum_match_reference_locp_py(golden, name):
    """g21: the objective of the reference's own locp.py WITH its input_nullspace term (locp.py:70-71, 258-261; vector and matrix)
    at ten seeded points and at the optimum; the oracle's solve (epigraph QP for a vector; kink / Newton cases for a matrix)
    reproduces the stored optimum and passes the duality certificate."""
    from oracle import locp as olocp
    g = golden('g21_locp_nullspace')
    case, ns = qp_cases.nullspace_case(name)
    pts = qp_cases.g14_points('nullspace_' + name, case)
    ws = [qp_cases.g14_pack(case, *p) for p in pts] + [g[name + '_wopt']]
    Jr, Rr = g[name + '_J'], g[name + '_res']
    for i, w in enumerate(ws):
        J, res, qp = qp_cases.g14_oracle_values(case, w)
        J += olocp.nullspace_term(qp, ns, w)
        np.testing.assert_allclose(J, Jr[i], rtol=1e-12, atol=1e-12 * max(1.0, abs(Jr[i])))
        np.testing.assert_allclose(res, Rr[i], rtol=0, atol=1e-12 * max(1.0, np.abs(Rr[i]).max()))
    w, J, info = olocp.solve_with_nullspace(qp, ns)
    assert J == pytest.approx(float(g[name + '_Jopt']), rel=1e-9)
    assert J == pytest.approx(float(Jr[-1]), rel=1e-9)                                    # the reference's own value at that point
    cert = olocp.nullspace_certificate(qp, ns, w, info['mu'])
    assert cert['mu_norm'] <= 1 + 1e-9 and abs(cert['gap']) <= 1e-9 and abs(cert['inner_dJ']) <= 1e-7, cert
    kink = name.endswith('kink')
    assert (olocp.nullspace_term(qp, ns, w) <= 1e-7) == kink and (cert['mu_norm'] < 0.5) == kink
    # a seeded feasible perturbation along the inputs never does better (convexity: a local check is a global one)
    N, n, m = qp.N, qp.n, qp.m
    rng = np.random.default_rng(5)
    Ad, Bd, dd = case['Ad'], case['Bd'], case['dd']
    x, u, sl = olocp.split(qp, w)
    for _ in range(5):
        u2 = np.clip(u + 1e-3 * rng.standard_normal(u.shape), 0.0, 800.0)
        x2 = [np.asarray(case['x0'], dtype=float)]
        for k in range(N):
            x2.append(Ad[k] @ x2[-1] + Bd[k] @ u2[k] + dd[k])
        x2 = np.array(x2)
        s2 = np.maximum(0.0, np.max(np.abs(case['x_scale'] * (x2 - case['xk'])), axis=1) - case['delta'])
        w2 = qp_cases.g14_pack(case, x2, u2, s2)
        if (qp.G @ w2 - qp.h).max() <= 1e-12:
            assert olocp.objective(qp, w2) + olocp.nullspace_term(qp, ns, w2) >= J - 1e-9


@pytest.mark.parametrize('tag', ['discrete', 'be', 'fe'])
def test_ssm_oracle_on_the_shipped_model(golden, tag):
    """oracle.ssm on the reference's shipped Diamond SSM model and recorded inputs (golden g17: the reference's own
    module test, examples/hardware/diamond_SSM.py:21-80): 1002-step open-loop rollout and its MSE against the
    measured tip trajectory."""
    from oracle import ssm as ossm
    g = golden('g17_ssm_hardware')
    model = ossm.make_model(6, 4, int(g['params_ROM_order'].item()), int(g['params_SSM_order'].item()), g['model_r_coeff'], g['model_B'],
                            g['model_w_coeff'], g['model_v_coeff'], g['z_eq'], rd_coeff=g['model_rd_coeff'], Bd=g['model_Bd'])
    kw = dict(discrete=dict(method='be', discrete=True), be=dict(method='be'), fe=dict(method='fe'))[tag]
    p, z = ossm.rollout(model, np.zeros(6), g['u_interp'], float(g['dt']), **kw)
    close(p, g[tag + '_p'], 1e-9)
    close(z, g[tag + '_z'], 1e-9)
    err = g['z_true_qv'] - g[tag + '_z'][:-1]
    assert np.linalg.norm(np.linalg.norm(err, axis=1)) ** 2 / err.shape[0] == pytest.approx(float(g[tag + '_mse']), rel=1e-12)


def test_pod_oracle_on_the_shipped_model(golden):
    """oracle.pod on the reference's shipped Diamond basis and rest state (golden g18)."""
    from oracle import pod as opod
    g = golden('g18_pod_shipped')
    U, q_ref, v_ref = g['U'], g['q_ref'], g['v_ref']
    n_f, r = U.shape
    rng = np.random.default_rng(int(g['seed']))
    Xq = q_ref + 5 * rng.standard_normal((5, n_f))
    Xv = rng.standard_normal((5, n_f))
    close(opod.project(U, q_ref, g['rest'][None])[0], g['proj_rest'], 1e-12)
    close(opod.project(U, q_ref, Xq), g['proj_q'], 1e-12)
    want = np.concatenate((opod.project(U, v_ref, Xv), opod.project(U, q_ref, Xq)), axis=1)
    close(want, g['proj_x'], 1e-12)
    M = np.random.default_rng(int(g['seed']) + 1).standard_normal((n_f, n_f))
    close(opod.reduce_matrix(U, M), g['UMU'], 1e-11)
    assert np.abs(U.T @ U - np.eye(r)).max() < 1e-13


@pytest.mark.parametrize('name', ['blobs', 'cloud'])
def test_g19_snapshot_preprocessing_and_kmeans_restatement(name):
    """oracle.pod.process_snapshots (incl. the numpy restatement of sklearn's KMeans that the reference calls) against the
    imported reference's outputs."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g19_preprocess.npz'))
    S, k = g[name], int(g[name + '_k'])
    assert np.abs(opod.process_snapshots(S, ['normalize'], {}) - g[name + '_normalize']).max() == 0.0
    assert np.abs(opod.process_snapshots(S, ['substract_mean'], {}) - g[name + '_mean']).max() == 0.0
    assert np.abs(opod.process_snapshots(S, ['normalize', 'substract_mean'], {}) - g[name + '_both']).max() == 0.0
    assert np.abs(opod.process_snapshots(S, ['clustering'], dict(nbr_clusters=k)) - g[name + '_centroids']).max() <= 1e-13 * np.abs(S).max()
    assert np.abs(opod.process_snapshots(S, ['normalize', 'substract_mean', 'clustering'], dict(nbr_clusters=k)) - g[name + '_all']).max() <= 1e-13
