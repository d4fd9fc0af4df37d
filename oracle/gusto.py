"""Oracle: GuSTO sequential-convex-programming outer loop (test infrastructure only).

Restates sofacontrol/scp/gusto.py:283-487 for a prediscretised nn-TPWL
model (oracle/tpwl.py dict) with the QP of oracle/locp.py.
"""
import numpy as np

from . import locp as olocp
from . import tpwl as otpwl

DEFAULTS = dict(delta0=1e4, omega0=1.0, rho=0.1, beta_fail=0.5, gamma_fail=5.0, epsilon=0.01,
                omega_max=1e10, max_gusto_iters=500, convg_thresh=0.1)  # gusto.py:12-22


def is_in_trust_region(x, xk, xs, delta, eps):
    """gusto.py:174-183."""
    md = np.max(np.linalg.norm(xs * (x - xk), np.inf, axis=1))
    return (md, False) if md - delta > eps else (0.0, True)


def is_converged(x, xk, xs, N, thresh):
    """gusto.py:150-161."""
    n = x.shape[1]
    dsol = (1. / N) * (1. / n) * np.sum(np.linalg.norm(xs * (x - xk), axis=1))
    return dsol, dsol <= thresh


def state_violation(X, x):
    """gusto.py:185-201 with utils.py:394-398 -- all rows k=0..N are checked."""
    if X is None:
        return 0.0
    A, b = X
    return float(max(np.linalg.norm(np.maximum(A @ x[i] - b, 0)) for i in range(x.shape[0])))


def compute_accuracy(model, x, u, xk, uk, J, dt, fs):
    """gusto.py:203-223 with continuous-time nearest-point dynamics (scp/models/tpwl.py:32-50);
    `model` may also be a callable (x, u) -> (f, A, B) (TemplateModel.get_continuous_dynamics)."""
    cont = model if callable(model) else (lambda xx, uu: otpwl.continuous_dynamics(model, xx, uu))
    err = approx = 0.0
    for i in range(x.shape[0] - 1):
        fk, Ak, Bk = cont(xk[i], uk[i])
        f, _, _ = cont(x[i], u[i])
        fa = fk + Ak @ (x[i] - xk[i]) + Bk @ (u[i] - uk[i])
        err += dt * np.linalg.norm(fs * (f - fa), 2)
        approx += dt * np.linalg.norm(fs * fa, 2)
    return err / (J + approx)


def traj_dynamics(model, Ad, Bd, dd, x):
    """gusto.py:225-238 -- table gather at the nearest point of x_k, k=0..N-1."""
    idx = otpwl.nearest_points(model, x[:-1])
    return Ad[idx], Bd[idx], dd[idx], idx


def solve(model, Ad, Bd, dd, H, N, dt, Qz, R, x0, u_init, x_init, z=None, u_des=None, Qzf=None,
          zf=None, U=None, X=None, Xf=None, dU=None, x_char=None, f_char=None, qp_solver=None,
          **kw):
    """One GuSTO.solve call.  Returns xopt, uopt, zopt and a per-iteration trace of
    (J, delta, omega, rho_k, accepted, tr_ok) tuples."""
    def get_traj(xk, uk):
        return traj_dynamics(model, Ad, Bd, dd, xk)[:3]
    return _loop(get_traj, model, H, N, dt, Qz, R, x0, u_init, x_init, z, u_des, Qzf, zf, U, X, Xf, dU, x_char,
                 f_char, qp_solver, kw)


def solve_generic(dyn_d, dyn_c, H, N, dt, Qz, R, x0, u_init, x_init, z=None, u_des=None, Qzf=None, zf=None,
                  U=None, X=None, Xf=None, dU=None, x_char=None, f_char=None, qp_solver=None, obs_lin=None, **kw):
    """The same loop for a generic TemplateModel: dyn_d(x, u) -> (A_d, B_d, d_d)
    (get_discrete_dynamics, gusto.py:225-238), dyn_c(x, u) -> (f, A, B) (get_continuous_dynamics);
    obs_lin(x) -> (H_d, c_d) for models with a nonlinear observer (gusto.py:240-251, 306-310, 467-471)."""
    def get_traj(xk, uk):
        A, B, d = zip(*[dyn_d(xk[i], uk[i]) for i in range(xk.shape[0] - 1)])
        return np.stack(A), np.stack(B), np.stack(d)
    return _loop(get_traj, dyn_c, H, N, dt, Qz, R, x0, u_init, x_init, z, u_des, Qzf, zf, U, X, Xf, dU, x_char,
                 f_char, qp_solver, kw, obs_lin)


def _loop(get_traj, model, H, N, dt, Qz, R, x0, u_init, x_init, z, u_des, Qzf, zf, U, X, Xf, dU, x_char, f_char,
          qp_solver, kw, obs_lin=None):
    par = dict(DEFAULTS); par.update(kw)

    def get_obs(xk):
        if obs_lin is None:
            return None, None
        Hs, cs = zip(*[obs_lin(xk[i]) for i in range(xk.shape[0])])
        return np.stack(Hs), np.stack(cs)
    n = x0.shape[0]
    xs = 1. / np.abs(x_char) if x_char is not None else np.ones(n)
    fs = 1. / np.abs(f_char) if f_char is not None else np.ones(n)
    stage_qp = isinstance(qp_solver, str) and qp_solver == 'riccati_ipm'
    cond_qp = isinstance(qp_solver, str) and qp_solver == 'condensed_ipm'
    if qp_solver is None:
        def qp_solver(qp):
            w, _, _ = olocp.solve_exact(qp)
            return w
    xk, uk = x_init.copy(), u_init.copy()
    A_k, B_k, d_k = get_traj(xk, uk)
    H_k, c_k = get_obs(xk)
    delta, omega = par['delta0'], par['omega0']
    new_solution = True
    J_prev = d_prev = o_prev = np.inf
    converged = False
    itr = 0
    trace = []
    warm_state, ipm_iters = None, []
    while itr <= par['max_gusto_iters'] and not converged and omega <= par['omega_max']:
        if stage_qp:
            # the stage-structured interior point (numpy statement of the kernel's algorithm) on the same QP data;
            # seconds instead of minutes at the BASELINE shapes.  Checked against solve_exact in tests/test_locp_oracle.py
            from . import riccati_ipm as ripm
            assert dU is None and obs_lin is None
            sp = ripm.Problem(N, H, Qz, R, A_k, B_k, d_k, x0, xk, delta, omega, z=z, u_des=u_des, Qzf=Qzf, zf=zf,
                              U=U, X=X, Xf=Xf, x_scale=xs)
            x_next, u_next, _, J, _ = ripm.solve(sp)
        elif cond_qp:
            # the condensed interior point (numpy statement of csrc/locp_cond.h) with the kernel's control flow: the QP
            # without its trust-region rows first; the full stage-structured solve only if that minimiser leaves the
            # trust region
            from . import riccati_ipm as ripm, condensed_ipm as cipm
            assert dU is None and obs_lin is None
            sp = ripm.Problem(N, H, Qz, R, A_k, B_k, d_k, x0, xk, delta, omega, z=z, u_des=u_des, Qzf=Qzf, zf=zf,
                              U=U, X=X, Xf=Xf, x_scale=xs)
            # (round 4) every QP after the first one the condensed path finished starts from that one's iterate
            x_next, u_next, J, info = cipm.solve(sp, warm=warm_state if par.get('warm_start_qp', True) else None)
            J += omega * max(0.0, np.max(np.abs(xs * (x0 - xk[0]))) - delta)       # s_0 (closed form)
            ipm_iters.append(info['iters'])
            if info['status'] == 'optimal':
                warm_state = info['final']
            if not (info['status'] == 'optimal' and info['inside']):
                x_next, u_next, _, J, _ = ripm.solve(sp)
        else:
            qp = olocp.build_qp(N, H, Qz, R, A_k, B_k, d_k, x0, xk, delta, omega, z=z, u_des=u_des,
                                Qzf=Qzf, zf=zf, U=U, X=X, Xf=Xf, dU=dU, x_scale=xs, Hd=H_k, cd=c_k)
            if par.get('input_nullspace') is not None:
                # locp.py:258-261: the objective (and with it the J of the accuracy ratio, gusto.py:405-420) carries the term
                w, J, _ = olocp.solve_with_nullspace(qp, par['input_nullspace'])
            else:
                w = qp_solver(qp)
                J = olocp.objective(qp, w)
            x_next, u_next, _ = olocp.split(qp, w)
        new_solution = False
        rho_k = -1.0
        e_tr, tr_ok = is_in_trust_region(x_next, xk, xs, delta, par['epsilon'])
        d_cur, o_cur = delta, omega
        if tr_ok:
            rho_k = compute_accuracy(model, x_next, u_next, xk, uk, J, dt, fs)
            if rho_k > par['rho'] and itr != 1:
                delta = par['beta_fail'] * delta
            else:
                if d_prev == delta and o_prev == omega and J_prev <= J:
                    delta = par['beta_fail'] * delta
                d_prev, J_prev, o_prev = delta, J, omega
                viol = state_violation(X, x_next)
                X_ok = not (viol > par['epsilon'])
                if not X_ok:
                    omega = par['gamma_fail'] * omega
                _, converged = is_converged(x_next, xk, xs, N, par['convg_thresh'])
                if not X_ok:
                    converged = False
                new_solution = True
        else:
            omega = par['gamma_fail'] * omega
        itr += 1
        trace.append((J, d_cur, o_cur, rho_k, new_solution, tr_ok))
        if new_solution:
            xk, uk = x_next.copy(), u_next.copy()
            if par['max_gusto_iters'] >= 1:
                A_k, B_k, d_k = get_traj(xk, uk)
                H_k, c_k = get_obs(xk)
    zopt = (H @ xk.T).T
    return xk, uk, zopt, trace
