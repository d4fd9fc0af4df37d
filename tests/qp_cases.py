"""Seeded LOCP test problems shared by the CPU (oracle) and GPU (HIP) parity tests."""
import numpy as np

from oracle import tpwl as otpwl, gusto as ogusto


def make_case(r=4, m=3, P=7, N=12, seed=30, q_scale=0.05, delta=1e4, omega=1.0, use_U=True, use_X=True,
              amp=0.15, dt=0.05, x0_scale=1e-4, u_max=800.0, xk_input=0.0, terminal=False, x_box=1.0):
    model = otpwl.synthetic_model(r, m, P, seed=seed)
    model['q'] = model['q'] * q_scale
    Ad, Bd, dd = otpwl.pre_discretize(model, dt, 'zoh')
    H = otpwl.synthetic_output_matrix(r, 6, seed + 1)
    rng = np.random.default_rng(seed + 2)
    n = 2 * r
    x0 = x0_scale * rng.standard_normal(n)
    u_init = xk_input * np.ones((N, m))
    xk = otpwl.rollout(model, Ad, Bd, dd, x0, u_init)
    A_k, B_k, d_k, idx = ogusto.traj_dynamics(model, Ad, Bd, dd, xk)
    Qz = np.diag([0, 0, 0, 100., 100., 0])
    R = 1e-5 * np.eye(m)
    th = np.linspace(0, 1.5, N + 1)
    z = np.zeros((N + 1, 6))
    z[:, 3] = -amp * np.sin(th)
    z[:, 4] = 0.5 * amp * np.sin(2 * th)
    xc, fc = otpwl.characteristic_vals(model)
    UA = np.kron(np.eye(m), np.array([[1.], [-1.]]))
    Ub = np.tile([u_max, 0.], m)
    Hz = np.zeros((2, 6)); Hz[0, 3] = 1; Hz[1, 4] = 1
    Hx = Hz @ H
    X = (np.vstack([-Hx, Hx]), x_box * np.array([0.02, 0.02, 0.04, 0.03])) if use_X else None
    case = dict(N=N, H=H, Qz=Qz, R=R, Ad=A_k, Bd=B_k, dd=d_k, x0=x0, xk=xk, delta=delta, omega=omega, z=z,
                U=(UA, Ub) if use_U else None, X=X, x_scale=1. / np.abs(xc))
    if terminal:
        case['Qzf'] = 10 * Qz
        case['zf'] = z[-1]
    extra = dict(model=model, Ad_tab=Ad, Bd_tab=Bd, dd_tab=dd, x_char=xc, f_char=fc, dt=dt, u_init=u_init, idx=idx)
    return case, extra


CASES = {
    'box_X': dict(),
    'free': dict(use_U=False, use_X=False),
    'box_only_tr_loose': dict(use_X=False, seed=31),
    'tr_active_small_delta': dict(use_X=False, seed=32, delta=2e-3, omega=1.0),
    'tr_active_big_omega': dict(seed=33, delta=5e-3, omega=1e4),
    'tr_tiny_delta_huge_omega': dict(use_X=False, seed=34, delta=1e-5, omega=1e8),
    'r5_N20': dict(r=5, m=4, P=9, N=20, seed=3, use_X=False, delta=1e-2, omega=100.0),
    'terminal_cost': dict(seed=35, use_X=False, terminal=True),
    'warm_centre': dict(seed=36, xk_input=60.0, delta=0.05, omega=10.0),
}


# ---------------------------------------------------------------------------------------------------------------
# g14: the QP *statement* pinned to the reference's own locp.py (objective 218-263, constraints 265-342), evaluated
# through the cvxpy stand-in of tests/golden/_cvxpy_eval.py.  The inputs below are regenerated from seeds by the
# golden generator (reference side) and by tests/test_oracle_golden.py (oracle side); only outputs are stored.
G14_CASES = {k: v for k, v in CASES.items() if k != 'terminal_cost'}   # Qzf at n_x != n_z: locp.py:252 is ill-formed there
G14_CASES.update({
    # n_x == n_z (r = 3): the only shape for which locp.py:252 (`x[N*n_z:]`) and 330 (`x[n_z:]`) type-check
    'qzf_r3': dict(r=3, m=2, P=5, N=9, seed=40, use_X=True, terminal=True),
    'xf_r3': dict(r=3, m=2, P=5, N=9, seed=41, use_X=False, extra=('Xf',)),
    'dU': dict(seed=42, use_X=False, extra=('dU',)),
    'dU_X_Xf': dict(seed=43, extra=('dU', 'Xf')),
    'u_des': dict(seed=44, extra=('u_des',)),
    'nlobs_r3': dict(r=3, m=2, P=5, N=9, seed=45, use_X=True, x_box=5.0, extra=('nlobs',)),
    'nlobs_free_r3': dict(r=3, m=2, P=5, N=9, seed=46, use_X=False, use_U=False, extra=('nlobs',)),
    'mpc_no_tr': dict(seed=47, extra=('no_tr',)),
    'mpc_no_tr_qzf_r3': dict(r=3, m=2, P=5, N=9, seed=48, terminal=True, extra=('no_tr', 'u_des')),
})


def g14_case(name):
    """Keyword arguments of oracle.locp.build_qp for a g14 case (all variants of locp.py's optional pieces)."""
    spec = dict(G14_CASES[name])
    extra = spec.pop('extra', ())
    case, info = make_case(**spec)
    n, m = case['Bd'][0].shape
    N = case['N']
    rng = np.random.default_rng(900 + spec.get('seed', 30))
    if 'Xf' in extra:
        A = rng.standard_normal((3, n))
        case['Xf'] = (A, np.abs(A @ case['xk'][-1]) + 0.05)
    if 'dU' in extra:
        case['dU'] = (np.kron(np.eye(m), np.array([[1.], [-1.]])), np.full(2 * m, 20.0))
    if 'u_des' in extra:
        case['u_des'] = rng.uniform(0, 50, (N, m))
    if 'nlobs' in extra:
        case['Hd'] = case['H'][None] + 0.1 * rng.standard_normal((N + 1,) + case['H'].shape)
        case['cd'] = 0.01 * rng.standard_normal((N + 1, case['H'].shape[0]))
    if 'no_tr' in extra:
        case['tr_active'] = False
    return case


# input_nullspace (locp.py:70-71, 258-261): a vector gives |sum_k v . u_k|, a matrix || M sum_k u_k ||_2; small weights leave the
# optimum where the norm is smooth, large ones pull it into the kink (M sum_k u_k = 0).  (kind, weight, seed of the base case)
NULLSPACE_CASES = {
    'vec_smooth': ('vec', 1e-3, 50),
    'vec_kink': ('vec', 10.0, 51),
    'mat_smooth': ('mat', 1e-3, 52),
    'mat_kink': ('mat', 1.0, 53),
}


def nullspace_case(name):
    """(case for oracle.locp.build_qp, input_nullspace) of a NULLSPACE_CASES entry: the g14 'u_des' shape with its own seed."""
    kind, weight, seed = NULLSPACE_CASES[name]
    case, info = make_case(seed=seed)
    n, m = case['Bd'][0].shape
    rng = np.random.default_rng(900 + seed)
    case['u_des'] = rng.uniform(0, 50, (case['N'], m))
    if kind == 'vec':
        ns = weight * rng.standard_normal(m)
    else:
        ns = np.zeros((2, m))                     # rows e_i - e_{i+1}: the inputs may stay equal and positive inside the kink
        for i in range(2):
            ns[i, i], ns[i, i + 1] = weight, -weight
    return case, ns


def g14_points(name, case, count=10):
    """Seeded evaluation points (x, u, s): random, around the trust-region centre, with nonnegative slacks."""
    N = case['N']
    n, m = case['Bd'][0].shape
    rng = np.random.default_rng(7000 + sum(map(ord, name)))
    pts = []
    for i in range(count):
        sc = 10.0 ** rng.uniform(-3, 1)
        x = case['xk'] + sc * rng.standard_normal((N + 1, n)) / case['x_scale']
        u = rng.uniform(-100, 900, (N, m))
        s = np.abs(rng.standard_normal(N + 1)) * sc
        pts.append((x, u, s))
    return pts


def g14_oracle_values(case, w):
    """The oracle's statement of the same quantities, in the reference's constraint order (locp.py:265-342):
    J; dynamics (287); per-stage trust-region norm residual (295) and slack positivity (297); U (303); dU (308);
    X (333 / 329); Xf (337); x_0 = x0 (340)."""
    from oracle import locp as olocp
    kw = dict(case)
    qp = olocp.build_qp(kw.pop('N'), kw.pop('H'), kw.pop('Qz'), kw.pop('R'), kw.pop('Ad'), kw.pop('Bd'), kw.pop('dd'),
                        kw.pop('x0'), kw.pop('xk'), kw.pop('delta'), kw.pop('omega'), **kw)
    N, n = qp.N, qp.n
    J = olocp.objective(qp, w)
    eq = qp.E @ w - qp.e
    g = qp.G @ w - qp.h
    parts = [eq[:N * n]]
    off = 0
    if qp.ns:
        tr = g[:(N + 1) * (2 * n + 1)].reshape(N + 1, 2 * n + 1)
        parts += [tr[:, :2 * n].max(axis=1), tr[:, 2 * n]]
        off = (N + 1) * (2 * n + 1)
    parts += [g[off:], eq[N * n:]]
    return J, np.concatenate(parts), qp


def g14_pack(case, x, u, s):
    return np.concatenate((x.ravel(), u.ravel(), s)) if case.get('tr_active', True) else np.concatenate((x.ravel(), u.ravel()))
