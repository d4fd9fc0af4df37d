"""GPU parity: the fused GuSTO kernel against the reference GuSTO loop (golden vectors g6: imported
reference class with the exact oracle QP injected for cvxpy) -- same iterates, same (J, delta, omega)
sequence; <= 1e-4 relative on trajectories."""
import io
import contextlib

import numpy as np
import pytest

from helpers import golden_problem, product_tpwl, Poly

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def setup(golden):
    g = golden('g6_gusto')
    model, U, q_ref, v_ref, Hf = golden_problem(4, 3, 7, 20, 30, q_scale=0.05)
    tp = product_tpwl(model, U, q_ref, v_ref, Hf)
    from sofacontrol_amd.scp.models.tpwl import TPWLGuSTO
    gm = TPWLGuSTO(tp)
    with contextlib.redirect_stdout(io.StringIO()):
        gm.pre_discretize(0.05)
    return g, gm


@pytest.mark.parametrize('tag', ['box', 'boxX', 'free'])
def test_gusto_node_first_solve_and_warm_resolve(golden, tag):
    from sofacontrol_amd.scp.standalone import GuSTOSolverNode
    g, gm = setup(golden)
    N, dt = 12, 0.05
    cons = {}
    if tag in ('box', 'boxX'):
        cons['U'] = Poly(g['U_A'], g['U_b'])
    if tag == 'boxX':
        cons['X'] = Poly(g['Xp_A'], g['Xp_b'])
    node = GuSTOSolverNode(gm, N, dt, g['Qz'], g['R'], np.zeros(8), t=g['t'], z=g['zt'], verbose=0,
                           warm_start=True, convg_thresh=1e-3, max_trace=64, **cons)
    xopt, uopt, zopt, topt = node.get_solution()
    ref_tr = g[tag + '_trace']
    assert int(node.gusto.iters[0]) == ref_tr.shape[0]
    tr = node.gusto.trace[0, :ref_tr.shape[0], :3]
    np.testing.assert_allclose(tr, ref_tr, rtol=1e-6)
    assert rel(xopt, g[tag + '_xopt']) <= 1e-4 and rel(uopt, g[tag + '_uopt']) <= 1e-4
    assert rel(zopt, g[tag + '_zopt']) <= 1e-4
    np.testing.assert_allclose(node.get_target(0.37)[0], g['get_target_z'], rtol=0, atol=1e-13)
    # receding-horizon callback with the shifted warm start (scp/ros.py:109-114)
    node.gusto.max_gusto_iters = 500
    t, x2, u2, z2, _ = node.gusto_callback(2 * dt, g[tag + '_x0b'])
    assert int(node.gusto.iters[0]) == g[tag + '_trace2'].shape[0]
    assert rel(x2, g[tag + '_xopt2']) <= 1e-4 and rel(u2, g[tag + '_uopt2']) <= 1e-4
    np.testing.assert_allclose(t, 2 * dt + dt * np.arange(N + 1))
    if tag == 'box':
        # the same request through the GuSTOsrv wire format (flat float64 lists, GuSTOsrv.srv / scp/ros.py:94-127)
        from sofacontrol_amd.scp.standalone import GuSTOsrvRequest
        node2 = GuSTOSolverNode(gm, N, dt, g['Qz'], g['R'], np.zeros(8), t=g['t'], z=g['zt'], verbose=0,
                                warm_start=True, convg_thresh=1e-3, max_trace=64, **cons)
        node2.gusto.max_gusto_iters = 500
        resp = node2.gusto_service(GuSTOsrvRequest(2 * dt, g[tag + '_x0b']))
        assert isinstance(resp.xopt, list) and len(resp.xopt) == (N + 1) * 8 and len(resp.uopt) == N * 3
        np.testing.assert_array_equal(np.array(resp.xopt).reshape(N + 1, 8), x2)
        np.testing.assert_array_equal(np.array(resp.uopt).reshape(N, 3), u2)
        np.testing.assert_array_equal(np.array(resp.zopt).reshape(N + 1, -1), z2)
        np.testing.assert_array_equal(np.array(resp.t), t)
        assert resp.solve_time >= 0.0


def test_gusto_helpers_match_reference(golden):
    from sofacontrol_amd.scp.gusto import GuSTO
    g, gm = setup(golden)
    N, dt = 12, 0.05
    x0 = np.zeros(8)
    u_init = np.zeros((N, 3))
    x_init, _ = gm.rollout(x0, u_init, dt)
    gu = GuSTO(gm, N, dt, g['Qz'], g['R'], x0, u_init, x_init, z=None, x_char=g['x_char'], f_char=g['f_char'],
               convg_thresh=1e-3, max_gusto_iters=0, X=Poly(g['Xp_A'], g['Xp_b']))
    gu.x_k, gu.u_k = g['h_xk'], g['h_uk']
    x, u = g['h_x'], g['h_u']
    np.testing.assert_allclose([gu.is_in_trust_region(x, d)[0] for d in (1e-3, 1e-1, 10.)], g['h_tr'], rtol=1e-12)
    np.testing.assert_allclose(gu.is_converged(x, u)[0], g['h_conv'], rtol=1e-12)
    np.testing.assert_allclose(gu.compute_accuracy(x, u, 3.7), g['h_rho'], rtol=1e-9)
    np.testing.assert_allclose(gu.state_constraints_violated(10 * x)[0], g['h_viol'], rtol=1e-12)


def test_gusto_batch_equals_single(golden):
    """Independent rollouts in one launch give the same answer as one-at-a-time solves."""
    from sofacontrol_amd.scp.gusto import GuSTO
    g, gm = setup(golden)
    N, dt, B = 12, 0.05, 5
    rng = np.random.default_rng(3)
    x0 = 1e-3 * rng.standard_normal((B, 8))
    u_init = np.zeros((B, N, 3))
    x_init, _ = gm.rollout(x0, u_init, dt)
    from scipy.interpolate import interp1d
    zi = interp1d(g['t'], g['zt'], axis=0)
    z = np.stack([zi(0.1 * b + dt * np.arange(N + 1)) for b in range(B)])
    kw = dict(x_char=g['x_char'], f_char=g['f_char'], convg_thresh=1e-3, U=Poly(g['U_A'], g['U_b']))
    gb = GuSTO(gm, N, dt, g['Qz'], g['R'], x0, u_init, x_init, z=z, batch=B, **kw)
    for b in range(B):
        g1 = GuSTO(gm, N, dt, g['Qz'], g['R'], x0[b], u_init[b], x_init[b], z=z[b], **kw)
        np.testing.assert_array_equal(g1.xopt, gb.xopt[b])
        np.testing.assert_array_equal(g1.uopt, gb.uopt[b])
        assert int(g1.iters[0]) == int(gb.iters[b])


def test_gusto_dispatch_order_does_not_change_results(golden):
    """More rollouts than CUs: from the second solve of a plan on, workgroups take the rollouts longest-first (by the
    iteration counts of the previous solve, `lpt_order_kernel`).  Same inputs -> bit-identical outputs."""
    from sofacontrol_amd.scp.gusto import GuSTO
    g, gm = setup(golden)
    N, dt, B = 12, 0.05, 300
    rng = np.random.default_rng(5)
    x0 = 1e-3 * rng.standard_normal((B, 8)) * rng.uniform(0.1, 30.0, (B, 1))
    u_init = np.zeros((B, N, 3))
    x_init, _ = gm.rollout(x0, u_init, dt)
    from scipy.interpolate import interp1d
    zi = interp1d(g['t'], g['zt'], axis=0)
    z = np.stack([zi(0.003 * b + dt * np.arange(N + 1)) for b in range(B)])
    kw = dict(x_char=g['x_char'], f_char=g['f_char'], convg_thresh=1e-3, U=Poly(g['U_A'], g['U_b']))
    gb = GuSTO(gm, N, dt, g['Qz'], g['R'], x0, u_init, x_init, z=z, batch=B, **kw)    # first solve: identity order
    first = (gb.xopt.copy(), gb.uopt.copy(), gb.iters.copy(), gb.status.copy())
    assert len(np.unique(first[2])) > 1          # the key of the sort is not constant
    for _ in range(6):       # later solves: sorted order; repeated, since what this guards against was sporadic (a race in
        gb.solve_batch(x0, u_init, x_init, z=z)   # the LDS set-up showed as 1e-10 differences in later rounds of a launch)
        np.testing.assert_array_equal(gb.xopt, first[0])
        np.testing.assert_array_equal(gb.uopt, first[1])
        np.testing.assert_array_equal(gb.iters, first[2])
        np.testing.assert_array_equal(gb.status, first[3])


def test_gusto_r36_split_panel_vs_oracle():
    """The fused GuSTO kernel at n_x = 72 (r = 36, the reference's shipped Diamond basis size): split-panel QP
    path inside the persistent SCP kernel, against the restated loop around the exact QP oracle."""
    from oracle import tpwl as otpwl, gusto as ogusto, locp as olocp
    from sofacontrol_amd.scp.models.tpwl import TPWLGuSTO
    from sofacontrol_amd.scp.gusto import GuSTO
    r, m, P, N, dt = 36, 4, 8, 12, 0.05
    model, U, q_ref, v_ref, Hf = golden_problem(r, m, P, 40, 33, q_scale=0.2)
    tp = product_tpwl(model, U, q_ref, v_ref, Hf)
    gm = TPWLGuSTO(tp)
    with contextlib.redirect_stdout(io.StringIO()):
        gm.pre_discretize(dt)
    Ad, Bd, dd = np.stack(tp.A_d), np.stack(tp.B_d), np.stack(tp.d_d)
    Qz = np.diag([0, 0, 0, 100., 100., 0]); R = 1e-5 * np.eye(m)
    H = np.asarray(tp.H)
    th = np.linspace(0, 1.0, N + 1)
    z = np.zeros((N + 1, 6)); z[:, 3] = -0.01 * np.sin(th); z[:, 4] = 0.005 * np.sin(2 * th)
    UA = np.kron(np.eye(m), np.array([[1.], [-1.]])); Ub = np.tile([800., 0.], m)
    x0 = np.zeros(2 * r)
    u_init = np.zeros((N, m))
    x_init = otpwl.rollout(model, Ad, Bd, dd, x0, u_init)
    xc, fc = otpwl.characteristic_vals(model)
    g = GuSTO(gm, N, dt, Qz, R, x0, u_init, x_init, z=z, U=Poly(UA, Ub), x_char=xc, f_char=fc, convg_thresh=1e-3,
              max_gusto_iters=4, max_trace=16)
    assert g._fused
    xo, uo, zo, _ = g.get_solution()
    xe, ue, ze, tr = ogusto.solve(model, Ad, Bd, dd, H, N, dt, Qz, R, x0, u_init, x_init, z=z, U=(UA, Ub), x_char=xc,
                                  f_char=fc, convg_thresh=1e-3, max_gusto_iters=500)
    assert int(g.iters[0]) == len(tr)
    np.testing.assert_allclose(g.trace[0, :len(tr), 0], [t[0] for t in tr], rtol=1e-6)
    assert rel(xo, xe) <= 1e-4 and rel(uo, ue) <= 1e-4


def test_gusto_fused_terminal_cost_set_and_input_target(golden):
    """The fused kernel with Qzf / zf, a terminal set Xf and an input target u (gusto.py:54-56 arguments) against the
    restated loop."""
    from oracle import tpwl as otpwl, gusto as ogusto
    from sofacontrol_amd.scp.gusto import GuSTO
    g, gm = setup(golden)
    model, U_, q_ref, v_ref, Hf = golden_problem(4, 3, 7, 20, 30, q_scale=0.05)
    tp = gm.dyn_sys
    N, dt = 12, 0.05
    Ad, Bd, dd = np.stack(tp.A_d), np.stack(tp.B_d), np.stack(tp.d_d)
    H = np.asarray(tp.H)
    from scipy.interpolate import interp1d
    z = interp1d(g['t'], g['zt'], axis=0)(dt * np.arange(N + 1))
    x0 = np.zeros(8); u_init = np.zeros((N, 3))
    x_init = otpwl.rollout(model, Ad, Bd, dd, x0, u_init)
    rng = np.random.default_rng(12)
    u_tgt = rng.uniform(0, 20, (N, 3))
    Qzf = 5 * g['Qz']
    XfA, Xfb = g["Xp_A"][:2], 50.0 * np.abs(g["Xp_b"][:2]) + 1.0      # loose terminal box: row handling, inactive
    gu = GuSTO(gm, N, dt, g['Qz'], g['R'], x0, u_init, x_init, z=z, u=u_tgt, Qzf=Qzf, zf=z[-1], U=Poly(g['U_A'], g['U_b']),
               Xf=Poly(XfA, Xfb), x_char=g['x_char'], f_char=g['f_char'], convg_thresh=1e-3, max_trace=32, max_gusto_iters=3)
    assert gu._fused
    # this target makes the nearest-neighbour model hop between two regions for tens of iterations: compare the
    # first four (max_gusto_iters = 3 -> iterations 0..3, gusto.py:163-172) of a fresh solve
    gu.solve(x0, u_init, x_init, z=z, zf=z[-1], u=u_tgt)
    xo, uo, zo, _ = gu.get_solution()
    xe, ue, ze, tr = ogusto.solve(model, Ad, Bd, dd, H, N, dt, g['Qz'], g['R'], x0, u_init, x_init, z=z, u_des=u_tgt, Qzf=Qzf,
                                  zf=z[-1], U=(g['U_A'], g['U_b']), Xf=(XfA, Xfb), x_char=g['x_char'], f_char=g['f_char'],
                                  convg_thresh=1e-3, max_gusto_iters=3)
    assert int(gu.iters[0]) == len(tr) == 4
    np.testing.assert_allclose(gu.trace[0, :len(tr), 0], [t[0] for t in tr], rtol=1e-6)
    assert rel(xo, xe) <= 1e-4 and rel(uo, ue) <= 1e-4


def test_plan_tables_survive_other_time_steps(golden):
    """A GuSTO plan (dt = 0.05) keeps its own discrete tables: rollouts / Jacobians / an EKF at another time step on the
    same model object must not change its result (each dt has its own immutable device handle, tpwl.handle_for)."""
    from sofacontrol_amd.scp.gusto import GuSTO
    g, gm = setup(golden)
    tp = gm.dyn_sys
    N, dt = 12, 0.05
    x0 = 1e-3 * np.ones(8); u_init = np.zeros((N, 3))
    x_init, _ = gm.rollout(x0, u_init, dt)
    from scipy.interpolate import interp1d
    z = interp1d(g['t'], g['zt'], axis=0)(dt * np.arange(N + 1))
    gu = GuSTO(gm, N, dt, g['Qz'], g['R'], x0, u_init, x_init, z=z, U=Poly(g['U_A'], g['U_b']), x_char=g['x_char'],
               f_char=g['f_char'], convg_thresh=1e-3)
    x1, u1 = gu.xopt.copy(), gu.uopt.copy()
    tp.rollout(x0, np.zeros((5, 3)), 0.01)                       # another time step on the same model
    A1, _, _ = tp.get_jacobians(x0, dt=0.01)
    A5, _, _ = tp.get_jacobians(x0, dt=0.05)
    assert np.abs(A1 - A5).max() > 1e-3
    gu.solve(x0, u_init, x_init, z=z)
    np.testing.assert_array_equal(gu.xopt, x1)
    np.testing.assert_array_equal(gu.uopt, u1)


def test_gusto_with_input_rate_constraints(golden):
    """GuSTO(dU=...) (gusto.py:54-56): the rate rows couple the stages, so the solve runs the generic loop around the
    state-augmented device QP; same iterates as the restated loop."""
    from oracle import tpwl as otpwl, gusto as ogusto
    from sofacontrol_amd.scp.gusto import GuSTO
    g, gm = setup(golden)
    model, U_, q_ref, v_ref, Hf = golden_problem(4, 3, 7, 20, 30, q_scale=0.05)
    tp = gm.dyn_sys
    N, dt = 10, 0.05
    Ad, Bd, dd = np.stack(tp.A_d), np.stack(tp.B_d), np.stack(tp.d_d)
    H = np.asarray(tp.H)
    from scipy.interpolate import interp1d
    z = interp1d(g['t'], g['zt'], axis=0)(dt * np.arange(N + 1))
    x0 = np.zeros(8); u_init = np.zeros((N, 3))
    x_init = otpwl.rollout(model, Ad, Bd, dd, x0, u_init)
    dA = np.kron(np.eye(3), np.array([[1.], [-1.]])); db = np.full(6, 20.0)
    gu = GuSTO(gm, N, dt, g['Qz'], g['R'], x0, u_init, x_init, z=z, U=Poly(g['U_A'], g['U_b']), dU=Poly(dA, db),
               x_char=g['x_char'], f_char=g['f_char'], convg_thresh=1e-3, max_gusto_iters=3)
    assert not gu._fused
    gu.solve(x0, u_init, x_init, z=z)
    xo, uo, zo, _ = gu.get_solution()
    xe, ue, ze, tr = ogusto.solve(model, Ad, Bd, dd, H, N, dt, g['Qz'], g['R'], x0, u_init, x_init, z=z,
                                  U=(g['U_A'], g['U_b']), dU=(dA, db), x_char=g['x_char'], f_char=g['f_char'],
                                  convg_thresh=1e-3, max_gusto_iters=3)
    assert int(gu.iters[0]) == len(tr)
    assert rel(xo, xe) <= 1e-4 and rel(uo, ue) <= 1e-4
    assert np.abs(np.diff(uo, axis=0)).max() <= 20.0 * (1 + 1e-6)


@pytest.mark.parametrize('weight', [2e-4, 0.05])
def test_gusto_with_input_nullspace_term(golden, weight):
    """GuSTO(input_nullspace=v) (locp.py:70-71, 258-261; the driver's sketch: diamond_SSM.py:258-259): the norm couples every
    stage's input, so the solve runs the generic loop around LOCP's dual maximisation; the objective that enters the
    model-accuracy ratio carries the term.  Same iterates as the restated loop (oracle: epigraph QP per iteration).  The small
    weight leaves |sum_k v . u_k| > 0, the large one drives it to zero."""
    from oracle import tpwl as otpwl, gusto as ogusto
    from sofacontrol_amd.scp.gusto import GuSTO
    g, gm = setup(golden)
    model, U_, q_ref, v_ref, Hf = golden_problem(4, 3, 7, 20, 30, q_scale=0.05)
    tp = gm.dyn_sys
    N, dt = 10, 0.05
    Ad, Bd, dd = np.stack(tp.A_d), np.stack(tp.B_d), np.stack(tp.d_d)
    H = np.asarray(tp.H)
    from scipy.interpolate import interp1d
    z = interp1d(g['t'], g['zt'], axis=0)(dt * np.arange(N + 1))
    x0 = np.zeros(8); u_init = np.zeros((N, 3))
    x_init = otpwl.rollout(model, Ad, Bd, dd, x0, u_init)
    v = weight * np.array([0.6, -0.5, 0.62])
    gu = GuSTO(gm, N, dt, g['Qz'], g['R'], x0, u_init, x_init, z=z, U=Poly(g['U_A'], g['U_b']), input_nullspace=v,
               x_char=g['x_char'], f_char=g['f_char'], convg_thresh=1e-3, max_gusto_iters=3)
    assert not gu._fused
    gu.solve(x0, u_init, x_init, z=z)
    xo, uo, zo, _ = gu.get_solution()
    xe, ue, ze, tr = ogusto.solve(model, Ad, Bd, dd, H, N, dt, g['Qz'], g['R'], x0, u_init, x_init, z=z,
                                  U=(g['U_A'], g['U_b']), x_char=g['x_char'], f_char=g['f_char'],
                                  convg_thresh=1e-3, max_gusto_iters=3, input_nullspace=v)
    assert int(gu.iters[0]) == len(tr)
    assert rel(xo, xe) <= 1e-4 and rel(uo, ue) <= 1e-4
    term = abs(float(v @ uo.sum(axis=0)))
    # and the term did something: the solve without it ends elsewhere
    x2, u2, _, _ = ogusto.solve(model, Ad, Bd, dd, H, N, dt, g['Qz'], g['R'], x0, u_init, x_init, z=z, U=(g['U_A'], g['U_b']),
                                x_char=g['x_char'], f_char=g['f_char'], convg_thresh=1e-3, max_gusto_iters=3)
    assert abs(float(v @ u2.sum(axis=0))) > term + 1e-6
    assert (term <= 1e-6) == (weight > 0.01), term
