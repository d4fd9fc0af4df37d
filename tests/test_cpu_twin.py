"""CPU: the native CPU twin (oracle/csrc/sofacontrol_cpu.cpp -- the cpu_baseline of bench.py) against the numpy oracle:
projection, the stage-structured interior point on the seeded QPs, the GuSTO loop on a small TPWL model."""
import numpy as np
import pytest

from oracle import condensed_ipm as cipm, cpu_twin, gusto as ogusto, locp as olocp, pod as opod, riccati_ipm as ripm, tpwl as otpwl
from qp_cases import CASES, make_case


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def test_projection_matches_numpy():
    rng = np.random.default_rng(0)
    U, _ = np.linalg.qr(rng.standard_normal((300, 8)))
    ref = rng.uniform(-100, 100, 300)
    X = ref + rng.standard_normal((37, 300))
    for threads in (1, 3):
        np.testing.assert_allclose(cpu_twin.project(U, ref, X, threads=threads), opod.project(U, ref, X), rtol=0, atol=1e-12)


@pytest.mark.parametrize('name', list(CASES))
def test_locp_matches_exact_solver(name):
    case, _ = make_case(**CASES[name])
    kw = dict(case)
    args = [kw.pop(k) for k in ('N', 'H', 'Qz', 'R', 'Ad', 'Bd', 'dd', 'x0', 'xk', 'delta', 'omega')]
    qp = olocp.build_qp(*args, **kw)
    w, _, _ = olocp.solve_exact(qp)
    xe, ue, se = olocp.split(qp, w)
    x, u, s, J, info = cpu_twin.locp_solve(*args, **kw)
    assert info['status'] == 0
    assert rel(x, xe) <= 1e-4 and rel(u, ue) <= 1e-4
    assert abs(J - olocp.objective(qp, w)) <= 1e-7 * max(1.0, abs(olocp.objective(qp, w)))


@pytest.mark.parametrize('name', list(CASES))
def test_condensed_twin_matches_numpy_statement(name):
    """algo='condensed' (the device kernel's control flow: condensed interior point of the QP without its trust-region rows,
    Riccati interior point of the full QP when that minimiser leaves the trust region) against the numpy statements of the
    same two algorithms: same interior-point iteration count, iterates to 1e-7."""
    case, _ = make_case(**CASES[name])
    kw = dict(case)
    args = [kw.pop(k) for k in ('N', 'H', 'Qz', 'R', 'Ad', 'Bd', 'dd', 'x0', 'xk', 'delta', 'omega')]
    x, u, s, J, info = cpu_twin.locp_solve(*args, **kw, algo='condensed')
    assert info['status'] == 0
    N, H, Qz, R, Ad, Bd, dd, x0, xk, delta, omega = args
    p = ripm.Problem(N, H, Qz, R, Ad, Bd, dd, x0, xk, delta, omega, z=kw.get('z'), zf=kw.get('zf'), u_des=kw.get('u_des'),
                     Qzf=kw.get('Qzf'), U=kw.get('U'), X=kw.get('X'), Xf=kw.get('Xf'), x_scale=kw.get('x_scale'),
                     tr_active=kw.get('tr_active', True))
    xe, ue, Je, inf = cipm.solve(p)
    if inf['status'] == 'optimal' and inf['inside']:
        assert info['iters'] == inf['iters']
    else:                                   # trust region active: the full QP on the Riccati path
        xe, ue, se, Je, inf = ripm.solve(p)
        assert info['iters'] == inf['iters']
    assert rel(x, xe) <= 1e-7 and rel(u, ue) <= 1e-7


def test_gusto_loop_condensed_twin_matches_riccati_twin():
    """The two native algorithms drive the same GuSTO loop to the same iterates (what bench.py's cpu_baseline relies on)."""
    model = otpwl.synthetic_model(4, 3, 7, seed=30)
    model['q'] = model['q'] * 0.05
    dt, N = 0.05, 12
    Ad, Bd, dd = otpwl.pre_discretize(model, dt, 'zoh')
    H = otpwl.synthetic_output_matrix(4, 6, 31)
    Qz = np.diag([0, 0, 0, 100., 100., 0]); R = 1e-5 * np.eye(3)
    th = np.linspace(0, 1.5, N + 1)
    z = np.zeros((N + 1, 6)); z[:, 3] = -0.15 * np.sin(th); z[:, 4] = 0.075 * np.sin(2 * th)
    UA = np.kron(np.eye(3), np.array([[1.], [-1.]])); Ub = np.tile([800., 0.], 3)
    xc, fc = otpwl.characteristic_vals(model)
    rng = np.random.default_rng(2)
    B = 3
    x0 = 1e-3 * rng.standard_normal((B, 8))
    u_init = np.zeros((B, N, 3))
    x_init = np.stack([otpwl.rollout(model, Ad, Bd, dd, x0[b], u_init[b]) for b in range(B)])
    zb = np.stack([z * (1 + 0.2 * b) for b in range(B)])
    out = {}
    for algo in ('riccati', 'condensed'):
        out[algo] = cpu_twin.gusto_solve(model, Ad, Bd, dd, H, N, dt, Qz, R, x0, u_init, x_init, z=zb, U=(UA, Ub), x_char=xc, f_char=fc,
                                         convg_thresh=1e-3, max_gusto_iters=8, threads=2, max_trace=16, algo=algo)
    assert (out['riccati'][2] == out['condensed'][2]).all()
    assert rel(out['condensed'][0], out['riccati'][0]) <= 1e-7 and rel(out['condensed'][1], out['riccati'][1]) <= 1e-7
    k = int(out['riccati'][2].max())
    np.testing.assert_allclose(out['condensed'][3][:, :k, :3], out['riccati'][3][:, :k, :3], rtol=1e-7)


def test_gusto_loop_matches_numpy_loop():
    model = otpwl.synthetic_model(4, 3, 7, seed=30)
    model['q'] = model['q'] * 0.05
    dt, N = 0.05, 12
    Ad, Bd, dd = otpwl.pre_discretize(model, dt, 'zoh')
    H = otpwl.synthetic_output_matrix(4, 6, 31)
    Qz = np.diag([0, 0, 0, 100., 100., 0]); R = 1e-5 * np.eye(3)
    th = np.linspace(0, 1.5, N + 1)
    z = np.zeros((N + 1, 6)); z[:, 3] = -0.15 * np.sin(th); z[:, 4] = 0.075 * np.sin(2 * th)
    UA = np.kron(np.eye(3), np.array([[1.], [-1.]])); Ub = np.tile([800., 0.], 3)
    xc, fc = otpwl.characteristic_vals(model)
    rng = np.random.default_rng(2)
    B = 3
    x0 = 1e-3 * rng.standard_normal((B, 8))
    u_init = np.zeros((B, N, 3))
    x_init = np.stack([otpwl.rollout(model, Ad, Bd, dd, x0[b], u_init[b]) for b in range(B)])
    zb = np.stack([z * (1 + 0.2 * b) for b in range(B)])
    xo, uo, iters, trace = cpu_twin.gusto_solve(model, Ad, Bd, dd, H, N, dt, Qz, R, x0, u_init, x_init, z=zb, U=(UA, Ub), x_char=xc,
                                                f_char=fc, convg_thresh=1e-3, max_gusto_iters=8, threads=2, max_trace=16)
    for b in range(B):
        xe, ue, ze, tr = ogusto.solve(model, Ad, Bd, dd, H, N, dt, Qz, R, x0[b], u_init[b], x_init[b], z=zb[b], U=(UA, Ub), x_char=xc,
                                      f_char=fc, convg_thresh=1e-3, max_gusto_iters=8, qp_solver='riccati_ipm')
        assert int(iters[b]) == len(tr)
        np.testing.assert_allclose(trace[b, :len(tr), :3], np.array([t[:3] for t in tr]), rtol=1e-6)
        assert rel(xo[b], xe) <= 1e-4 and rel(uo[b], ue) <= 1e-4


def test_gusto_loop_warm_started_qps_follow_the_cold_started_loop():
    """Round 4: inside one GuSTO solve every QP after the first starts from the previous QP's minimiser and multipliers
    (oracle.gusto: warm_start_qp, default on; the twin and the lean kernels do the same).  The SCP loop it drives is the one the
    cold-started QPs drive -- same iteration counts, same (J, delta, omega) trace, trajectories to 1e-6 -- with fewer interior-point
    iterations, and the native twin follows the numpy statement."""
    model = otpwl.synthetic_model(4, 3, 7, seed=30)
    model['q'] = model['q'] * 0.05
    dt, N = 0.05, 12
    Ad, Bd, dd = otpwl.pre_discretize(model, dt, 'zoh')
    H = otpwl.synthetic_output_matrix(4, 6, 31)
    Qz = np.diag([0, 0, 0, 100., 100., 0]); R = 1e-5 * np.eye(3)
    th = np.linspace(0, 1.5, N + 1)
    z = np.zeros((N + 1, 6)); z[:, 3] = -0.15 * np.sin(th); z[:, 4] = 0.075 * np.sin(2 * th)
    UA = np.kron(np.eye(3), np.array([[1.], [-1.]])); Ub = np.tile([800., 0.], 3)
    xc, fc = otpwl.characteristic_vals(model)
    x0 = 1e-3 * np.random.default_rng(2).standard_normal(8)
    u_init = np.zeros((N, 3))
    x_init = otpwl.rollout(model, Ad, Bd, dd, x0, u_init)
    counts = {}
    orig = cipm.solve

    def counting(p, **kw):
        out = orig(p, **kw)
        counts[key].append(out[3]['iters'])
        return out
    res = {}
    cipm.solve = counting
    try:
        for key, flag in (('warm', True), ('cold', False)):
            counts[key] = []
            res[key] = ogusto.solve(model, Ad, Bd, dd, H, N, dt, Qz, R, x0, u_init, x_init, z=z, U=(UA, Ub), x_char=xc, f_char=fc,
                                    convg_thresh=1e-3, max_gusto_iters=8, qp_solver='condensed_ipm', warm_start_qp=flag)
    finally:
        cipm.solve = orig
    (xw, uw, _, tw), (xc_, uc_, _, tc) = res['warm'], res['cold']
    assert len(tw) == len(tc) >= 2
    np.testing.assert_allclose(np.array([t[:3] for t in tw]), np.array([t[:3] for t in tc]), rtol=1e-6)
    assert rel(xw, xc_) <= 1e-6 and rel(uw, uc_) <= 1e-6
    assert counts['warm'][0] == counts['cold'][0] and sum(counts['warm']) < sum(counts['cold'])
    xo, uo, iters, trace = cpu_twin.gusto_solve(model, Ad, Bd, dd, H, N, dt, Qz, R, x0[None], u_init[None], x_init[None], z=z[None], U=(UA, Ub),
                                                x_char=xc, f_char=fc, convg_thresh=1e-3, max_gusto_iters=8, threads=1, max_trace=16, algo='condensed')
    assert int(iters[0]) == len(tw) and rel(xo[0], xw) <= 1e-8 and rel(uo[0], uw) <= 1e-8


def test_ilqr_twin_tpwl_matches_numpy_oracle():
    """cpu_twin.ilqr_tpwl (the CPU number beside bench.py's `ilqr_diamond`) against oracle.lqr.ILQR -- the numpy restatement of
    sofacontrol/lqr/ilqr.py that the goldens g4 / g20 pin to the imported reference -- on a small nearest-point TPWL model: same
    iteration count, trajectory and gains; also with the four configuration switches off (lqr/config.py:6-9,31)."""
    from oracle import lqr as olqr
    from helpers import golden_problem
    import workloads as wl
    r, m, P, N, dt = 5, 4, 12, 10, 0.05
    model, U, q_ref, v_ref, Hf = golden_problem(r, m, P, 40, 5, q_scale=0.2)
    Ad, Bd, dd = wl.zoh_tables(dict(A_c=model['A_c'], B_c=model['B_c'], d_c=model['d_c']), dt)
    rng = np.random.default_rng(3)
    H = 0.3 * rng.standard_normal((6, 2 * r))
    z_ref = 0.01 * rng.standard_normal(6)
    Qz = np.diag([0., 0., 0., 100., 100., 10.]); R = 1e-3 * np.eye(m)
    th = np.linspace(0, 1.0, N + 1)
    zt = np.zeros((N + 1, 6)); zt[:, 3] = -0.02 * np.sin(th); zt[:, 4] = 0.01 * np.sin(2 * th)
    zt = zt + z_ref
    x0 = 1e-3 * rng.standard_normal((3, 2 * r))
    for switches in ({}, dict(include_input_var_constraint=False, do_linesearch=False, regularize=False, state_regularization=False)):
        x, u, K, cost, iters = cpu_twin.ilqr_tpwl(model, Ad, Bd, dd, H, z_ref, Qz, R, 10 * Qz, N, x0, np.stack([zt] * 3), threads=2, **switches)
        for b in range(3):
            o = olqr.ILQR(model, Ad, Bd, dd, H, z_ref, Qz, R, 10 * Qz, N)
            for k, v in switches.items():
                setattr(o.p, k, v)
            xo, uo, Ko = o.solve(x0[b], zt)
            # (all switches off: the iteration reaches a fixed point whose cost repeats EXACTLY in numpy -- the stopping rule
            # 0 <= J_old - J_new then depends on the last bit, and the twin's FMA-contracted sums may take extra, identical passes)
            if not switches:
                assert int(iters[b]) == len(o.trace) - 1, (switches, b, iters[b], len(o.trace) - 1)
            assert rel(x[b], xo) <= 1e-8 and rel(u[b], uo) <= 1e-7 and rel(K[b], Ko) <= 1e-6, (rel(x[b], xo), rel(u[b], uo), rel(K[b], Ko))
            assert abs(cost[b] - o.trace[-1][1]) <= 1e-9 * max(1.0, abs(o.trace[-1][1]))


@pytest.mark.parametrize('method,dt', [('fe', 0.01), ('be', 0.01), ('bil', 0.01)])
def test_ilqr_twin_ssm_matches_numpy_oracle(method, dt):
    """cpu_twin.ilqr_ssm (the CPU number beside bench.py's C3 entry) against oracle.lqr.ILQRGeneric over oracle.ssm (pinned to the
    imported reference by g10 / g11) on a small cubic SSM model at a well-conditioned step: same iteration count, trajectories."""
    from oracle import lqr as olqr, ssm as ossm
    n, m, N = 4, 3, 20
    model = ossm.synthetic(n, m, 3, 2, seed=7)
    Hc = model['W'][:, :n].copy()
    rng = np.random.default_rng(5)
    Q = np.diag([100., 100., 1., 1.]); R = np.eye(m)
    x0 = 0.05 * rng.standard_normal((2, n))
    th = np.linspace(0, 2 * np.pi, N + 1)
    zt = np.zeros((2, N + 1, n)); zt[:, :, 0] = 0.1 * np.sin(th); zt[1, :, 1] = 0.05 * (1 - np.cos(th))
    zt = zt + model['z_ref']
    x, u, K, cost, iters = cpu_twin.ilqr_ssm(n, m, 3, 2, model['R'], model['B'], model['W'], model['z_ref'], Hc, method, dt, Q, R, Q, N, x0, zt, threads=2)
    for b in range(2):
        o = olqr.ILQRGeneric(lambda xx, uu: ossm.jacobians(model, xx, uu, dt, method), lambda xx: ossm.observe(model, xx) + model['z_ref'],
                             Hc, n, m, Q, R, Q, N)
        xo, uo, Ko = o.solve(x0[b], zt[b])
        assert int(iters[b]) == len(o.trace) - 1, (b, iters[b], len(o.trace) - 1)
        assert rel(x[b], xo) <= 1e-8 and rel(u[b], uo) <= 1e-7, (rel(x[b], xo), rel(u[b], uo))
        assert abs(cost[b] - o.trace[-1][1]) <= 1e-9 * max(1.0, abs(o.trace[-1][1]))
