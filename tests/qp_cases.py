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
