"""GPU parity: Riccati recursions and iLQR against the golden vectors of the imported reference (g4)."""
import io
import contextlib

import numpy as np
import pytest

from helpers import golden_problem, product_tpwl

pytestmark = pytest.mark.gpu


def close(a, b, rtol):
    np.testing.assert_allclose(a, b, rtol=0, atol=rtol * max(1.0, float(np.abs(b).max())))


def setup(golden):
    g = golden('g4_riccati')
    model, U, q_ref, v_ref, Hf = golden_problem(5, 4, 9, 30, 20)
    tp = product_tpwl(model, U, q_ref, v_ref, Hf)
    with contextlib.redirect_stdout(io.StringIO()):
        tp.pre_discretize(0.05)
    return g, tp


def test_fixed_point_riccati_and_dare(golden):
    from sofacontrol_amd.lqr.lqr import solve_riccati, dare
    g, tp = setup(golden)
    L, P = solve_riccati(tp.A_d[2], tp.B_d[2], g['Q'], g['R'])
    close(L, g['sr_L'], 1e-9); close(P, g['sr_P'], 1e-9)
    K, P = dare(tp.A_d[2], tp.B_d[2], g['Q'], g['R'])
    close(K, g['dare_K'], 1e-8); close(P, g['dare_P'], 1e-8)


def test_traj_tracking_lqr(golden):
    from sofacontrol_amd.lqr.traj_tracking_lqr import TrajTrackingLQR
    from sofacontrol_amd.utils import QuadraticCost
    g, tp = setup(golden)

    class Target:
        t, x, u = g['tt_t'], g['tt_x'], g['tt_u']
    tt = TrajTrackingLQR(0.05, tp, QuadraticCost(Q=g['Q'], R=g['R']))
    xbar, ubar, K = tt.compute_policy(Target)
    close(xbar, g['tt_xbar'], 1e-12); close(ubar, g['tt_ubar'], 1e-12)
    close(K, g['tt_K'], 1e-9)
    K2, P = tt.perform_dlqr_recursion(Target)
    close(P, g['tt_P'], 1e-9)


@pytest.mark.parametrize('tag,N', [('c1', 10), ('n30', 30)])
def test_ilqr_full_solve(golden, tag, N):
    """Full ilqr_computation: same iteration count and trajectories as the reference (tolerance 1e-6:
    the line search compares cost ratios, trajectories agree to round-off until a decision flips)."""
    from sofacontrol_amd.lqr.ilqr import iLQR
    from sofacontrol_amd.utils import QuadraticCost
    g, tp = setup(golden)
    il = iLQR(0.05, tp, QuadraticCost(Q=g['Qz'], R=g['R'], Qf=g['Qf']), N)
    il.set_target(g[tag + '_z_target'])
    x, u, K = il.ilqr_computation(g[tag + '_x0'], g[tag + '_uw'])
    assert int(il.iters[0]) == int(g[tag + '_iters'])
    close(x, g[tag + '_sol_x'], 1e-6); close(u, g[tag + '_sol_u'], 1e-6); close(K, g[tag + '_sol_K'], 1e-6)
    x, u, K = il.ilqr_computation(g[tag + '_x0'])
    assert int(il.iters[0]) == int(g[tag + '_iters0'])
    close(x, g[tag + '_sol0_x'], 1e-6); close(u, g[tag + '_sol0_u'], 1e-6)
    # batched: two problems in one launch equal the single solves
    xb, ub, Kb = il.ilqr_computation(np.stack([g[tag + '_x0'], g[tag + '_x0']]))
    np.testing.assert_array_equal(xb[0], x); np.testing.assert_array_equal(xb[1], x)


@pytest.mark.parametrize('case', ['reference', 'include_input_var_constraint', 'do_linesearch', 'regularize', 'state_regularization', 'all_off'])
def test_ilqr_config_switches(golden, case):
    """The four switches of lqr/config.py:6-9, 31 as kernel parameters (silqr_params): golden g20 = the imported reference
    iLQR with each switch (and all of them) turned off -- same iteration counts, trajectories and gains."""
    from sofacontrol_amd.lqr.ilqr import iLQR
    from sofacontrol_amd.utils import QuadraticCost
    g4, tp = setup(golden)
    g = golden('g20_ilqr_switches')
    flags = ('include_input_var_constraint', 'do_linesearch', 'regularize', 'state_regularization')
    for warm in (True, False):
        il = iLQR(0.05, tp, QuadraticCost(Q=g['Qz'], R=g['R'], Qf=g['Qf']), int(g['N']))
        for f in flags:
            setattr(il.params, f, not (case == f or case == 'all_off'))
        il.set_target(g['z_target'])
        il.set_u_last(g['u_last'])
        x, u, K = il.ilqr_computation(g['x0'], g['uw'] if warm else None)
        key = case + ('_warm' if warm else '_cold')
        # all switches off: the problem is solved after ONE full Newton step; the reference's second and third iterations move
        # the cost by -4e-12 and +3e-12 (5965.533874068984 -> ...988 -> ...985) and its stopping rule `0 <= decrease < epsilon`
        # (ilqr.py:109-115) waits for the sign of that rounding noise: the count may differ by one, the result may not
        slack = 1 if case == 'all_off' else 0
        assert abs(int(il.iters[0]) - int(g[key + '_iters'])) <= slack, (key, int(il.iters[0]), int(g[key + '_iters']))
        close(x, g[key + '_x'], 1e-6); close(u, g[key + '_u'], 1e-6); close(K, g[key + '_K'], 1e-5)


@pytest.mark.parametrize('r,m', [(30, 4), (36, 4)])
def test_ilqr_diamond_sizes_vs_oracle(r, m):
    """iLQR at the Diamond state sizes (n_x = 60 and the shipped r = 36 basis, n_x = 72) against the oracle loop."""
    import io, contextlib
    from oracle import lqr as olqr, tpwl as otpwl
    from helpers import golden_problem, product_tpwl
    from sofacontrol_amd.lqr.ilqr import iLQR
    from sofacontrol_amd.utils import QuadraticCost
    P, N, dt = 8, 15, 0.05
    model, U, q_ref, v_ref, Hf = golden_problem(r, m, P, 40, 55, q_scale=0.2)
    tp = product_tpwl(model, U, q_ref, v_ref, Hf)
    with contextlib.redirect_stdout(io.StringIO()):
        tp.pre_discretize(dt)
    Ad, Bd, dd = np.stack(tp.A_d), np.stack(tp.B_d), np.stack(tp.d_d)
    H, z_ref = np.asarray(tp.H), np.asarray(tp.z_ref)
    Qz = np.diag([0., 0., 0., 100., 100., 10.]); R = 1e-3 * np.eye(m)
    th = np.linspace(0, 1.0, N + 1)
    zt = np.zeros((N + 1, 6)); zt[:, 3] = -0.02 * np.sin(th); zt[:, 4] = 0.01 * np.sin(2 * th)
    zt = zt + z_ref
    x0 = 1e-3 * np.random.default_rng(1).standard_normal(2 * r)
    il = iLQR(dt, tp, QuadraticCost(Q=Qz, R=R, Qf=10 * Qz), N)
    il.set_target(zt)
    x, u, K = il.ilqr_computation(x0)
    o = olqr.ILQR(model, Ad, Bd, dd, H, z_ref, Qz, R, 10 * Qz, N)
    xo, uo, Ko = o.solve(x0, zt)
    assert int(il.iters[0]) == len(o.trace) - 1
    close(x, xo, 1e-6); close(u, uo, 1e-6); close(K, Ko, 1e-5)


def test_baseline_config_c1_r5_horizon10():
    """BASELINE.json configs[0] -- Diamond TPWL ROM r = 5 (n_x = 10), LQR horizon 10: iLQR plan and the TV-LQR tracking
    gains around it, against the oracle (the numpy restatement of the reference's CPU path)."""
    import io, contextlib
    from oracle import lqr as olqr
    from helpers import golden_problem, product_tpwl
    from sofacontrol_amd.lqr.ilqr import iLQR
    from sofacontrol_amd.lqr.traj_tracking_lqr import TrajTrackingLQR
    from sofacontrol_amd.utils import QuadraticCost
    r, m, P, N, dt = 5, 4, 12, 10, 0.05
    model, U, q_ref, v_ref, Hf = golden_problem(r, m, P, 40, 5, q_scale=0.2)
    tp = product_tpwl(model, U, q_ref, v_ref, Hf)
    with contextlib.redirect_stdout(io.StringIO()):
        tp.pre_discretize(dt)
    Ad, Bd, dd = np.stack(tp.A_d), np.stack(tp.B_d), np.stack(tp.d_d)
    H, z_ref = np.asarray(tp.H), np.asarray(tp.z_ref)
    Qz = np.diag([0., 0., 0., 100., 100., 10.]); R = 1e-3 * np.eye(m)
    th = np.linspace(0, 1.0, N + 1)
    zt = np.zeros((N + 1, 6)); zt[:, 3] = -0.02 * np.sin(th); zt[:, 4] = 0.01 * np.sin(2 * th)
    zt = zt + z_ref
    x0 = 1e-3 * np.random.default_rng(5).standard_normal(2 * r)
    il = iLQR(dt, tp, QuadraticCost(Q=Qz, R=R, Qf=10 * Qz), N)
    il.set_target(zt)
    x, u, K = il.ilqr_computation(x0)
    o = olqr.ILQR(model, Ad, Bd, dd, H, z_ref, Qz, R, 10 * Qz, N)
    xo, uo, Ko = o.solve(x0, zt)
    assert int(il.iters[0]) == len(o.trace) - 1
    close(x, xo, 1e-8); close(u, uo, 1e-8); close(K, Ko, 1e-7)
    # TV-LQR around the plan (traj_tracking_lqr.py:18-48)
    Q = H.T @ Qz @ H + 1e-3 * np.eye(2 * r)

    class Target:
        t, x, u = dt * np.arange(N + 1), xo, np.vstack((uo, uo[-1:]))
    tt = TrajTrackingLQR(dt, tp, QuadraticCost(Q=Q, R=R))
    Kt, Pt = tt.perform_dlqr_recursion(Target)
    idx = [int(np.argmin(np.linalg.norm(model['q'] - xi[r:], axis=1))) for xi in np.asarray(Target.x)]
    Po = Q.copy()
    for k in range(len(Kt) - 1, -1, -1):
        A, B = Ad[idx[k]], Bd[idx[k]]
        Kk = -np.linalg.solve(R + B.T @ Po @ B, B.T @ Po @ A)
        close(Kt[k], Kk, 1e-8)
        Acl = A + B @ Kk
        Po = Q + Kk.T @ R @ Kk + Acl.T @ Po @ Acl


def _dare_case(n, m, rho, rank_q, seed):
    rng = np.random.default_rng(seed)
    V = rng.standard_normal((n, n))
    lam = rho * rng.uniform(0.3, 1.0, n)
    lam[0] = rho                                     # the slowest mode sits at |lambda| = rho
    A = np.real(V @ np.diag(lam) @ np.linalg.inv(V))
    B = rng.standard_normal((n, m))
    Cq = rng.standard_normal((rank_q, n))
    return A, B, Cq.T @ Cq, np.diag(rng.uniform(0.5, 2.0, m)) * 1e-2


@pytest.mark.parametrize('n,m,rho,rank_q', [(8, 2, 0.9, 8), (20, 3, 0.9999, 20), (60, 4, 0.999, 2), (60, 8, 1.02, 6),
                                             (72, 4, 0.99, 3), (33, 16, 0.95, 33)])
def test_dare_doubling_vs_scipy(n, m, rho, rank_q, monkeypatch):
    """sric_dare (structure-preserving doubling) against scipy.linalg.solve_discrete_are: lightly damped and unstable
    open loops, rank-deficient state cost, n_x = 72 (slots in HBM instead of LDS), n_u = 16; also the batched call and
    the forced HBM-slot path at a size that would fit LDS."""
    import scipy.linalg as sl
    from sofacontrol_amd.lqr.lqr import dare, dare_batch
    A, B, Q, R = _dare_case(n, m, rho, rank_q, 100 * n + m)
    K, P = dare(A, B, Q, R)
    Ps = sl.solve_discrete_are(A, B, Q, R)
    Ks = -np.linalg.solve(R + B.T @ Ps @ B, B.T @ Ps @ A)
    close(P, Ps, 1e-9); close(K, Ks, 1e-8)
    assert np.abs(np.linalg.eigvals(A + B @ K)).max() < 1.0
    A2, B2, _, _ = _dare_case(n, m, min(rho, 0.97), rank_q, 7 * n + m)
    Kb, Pb = dare_batch(np.stack([A, A2]), np.stack([B, B2]), Q, R)
    np.testing.assert_array_equal(Pb[0], P)
    close(Pb[1], sl.solve_discrete_are(A2, B2, Q, R), 1e-9)
    if n <= 60:
        monkeypatch.setenv('SRH_DARE_HBM_SLOTS', '1')
        K3, P3 = dare(A, B, Q, R)
        close(P3, Ps, 1e-9)


def test_dare_reports_failure():
    from sofacontrol_amd.lqr.lqr import dare
    A = np.diag([1.5, 0.5]); B = np.array([[0.0], [1.0]])        # the unstable mode is not controllable
    with pytest.raises(Exception):
        dare(A, B, np.eye(2), np.eye(1))
