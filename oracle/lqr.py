"""Oracle: Riccati recursions and iLQR (test infrastructure only).

Restates sofacontrol/lqr/lqr.py, sofacontrol/lqr/traj_tracking_lqr.py and
sofacontrol/lqr/ilqr.py (+ sofacontrol/lqr/config.py) in numpy float64, on a TPWL
model given as the dict of oracle/tpwl.py plus prediscretised tables.
"""
import numpy as np
import scipy.linalg

from . import tpwl as otpwl


def solve_riccati(A, B, Q, R, tol=1e-4, max_iter=100000):
    """sofacontrol/lqr/lqr.py:6-21 -- fixed-point DARE, stops at ||L - L_old||_F <= 1e-4."""
    n, m = B.shape
    P = np.zeros((n, n))
    L = np.linalg.solve(R + B.T @ P @ B, B.T @ P @ A)
    Lold = np.inf * np.ones((m, n))
    it = 0
    while np.linalg.norm(L - Lold) > tol and it < max_iter:
        Lold = L
        P = A.T @ P @ A - A.T @ P @ B @ np.linalg.inv(R + B.T @ P @ B) @ (B.T @ P @ A) + Q
        L = -np.linalg.solve(R + B.T @ P @ B, B.T @ P @ A)
        it += 1
    return L, P, it


def dare(A, B, Q, R):
    """sofacontrol/lqr/lqr.py:24-31."""
    P = scipy.linalg.solve_discrete_are(A, B, Q, R)
    K = -scipy.linalg.inv(B.T @ P @ B + R) @ (B.T @ P @ A)
    return K, P


def tvlqr(A, B, Q, R):
    """sofacontrol/lqr/traj_tracking_lqr.py:18-48 for given per-step (A_i, B_i), i=0..n-1 in
    *forward* time order; terminal P = Q.  Returns K (n, m, nx), P (n+1, nx, nx) forward order."""
    n = A.shape[0]
    P = [Q]
    K = []
    for i in reversed(range(n)):
        Ki = -np.linalg.solve(R + B[i].T @ P[-1] @ B[i], B[i].T @ P[-1] @ A[i])
        K.append(Ki)
        Acl = A[i] + B[i] @ Ki
        P.append(Q + Ki.T @ R @ Ki + Acl.T @ P[-1] @ Acl)
    return np.flip(np.asarray(K), axis=0), np.flip(np.asarray(P), axis=0)


class ILQRParams:
    """sofacontrol/lqr/config.py:1-31."""
    max_iter = 50
    epsilon = 0.1
    alpha0 = 1.
    alpha_scaling = 0.5
    improv_lb = 1e-4
    improv_ub = 100
    alpha_min = 5e-2
    counter_limit = 5
    rho0 = 0.
    drho0 = 0.
    rho_scaling = 1.5
    rho_increase_fp = 10.
    rho_max = 1e5
    rho_min = 1e-3
    include_input_var_constraint = True        # config.py:6
    do_linesearch = True                       # config.py:8
    regularize = True                          # config.py:9
    state_regularization = True                # config.py:31


class ILQR:
    """sofacontrol/lqr/ilqr.py:6-300 on a prediscretised nn-TPWL model; the four switches of config.py (input-variation cost,
    line search, regularisation, state / input regularisation) live in `self.p` (an instance copy of ILQRParams)."""

    def __init__(self, model, Ad, Bd, dd, H, z_ref, Q, R, Qf, N):
        self.model, self.Ad, self.Bd, self.dd = model, Ad, Bd, dd
        self.H, self.z_ref, self.Q, self.R, self.Qf, self.N = H, z_ref, Q, R, Qf, N
        self.n, self.m = Bd.shape[1], Bd.shape[2]
        self.p = ILQRParams()
        self.u_last = np.zeros(self.m)
        self.z_target = None
        self.trace = []

    def _z(self, x):
        """model.x_to_zfyf(x, zf=True)."""
        return self.H @ x + self.z_ref

    def _lin(self, x, u):
        """model.get_jacobians(x, u=u, dt=dt) (ilqr.py:155)."""
        i = otpwl.nearest_point(self.model, x)
        return self.Ad[i], self.Bd[i], self.dd[i]

    def _cost_terms(self, x, u, t, u_prev):
        z = self._z(x)
        dz = z - self.z_target[t]
        du = u - u_prev
        return .5 * dz @ self.Q @ dz + .5 * du @ self.R @ du

    def forward_pass(self, x_prev, u_prev, alpha=1., K=None, k=None):
        """ilqr.py:117-162."""
        N, n, m = self.N, self.n, self.m
        x = np.zeros((N + 1, n)); u = np.zeros((N, m))
        A = np.zeros((N, n, n)); B = np.zeros((N, n, m)); d = np.zeros((N, n))
        x[0] = x_prev[0]
        if K is None:
            K = np.zeros((N, m, n))
        if k is None:
            k = np.zeros((N, m))
        cost = 0.
        for t in range(N):
            u[t] = u_prev[t] + alpha * k[t] + K[t] @ (x[t] - x_prev[t])
            if self.p.include_input_var_constraint:
                cost += self._cost_terms(x[t], u[t], t, self.u_last if t == 0 else u[t - 1])
            else:
                cost += self._cost_terms(x[t], u[t], t, np.zeros(m))                 # ilqr.py:150-152: u' R u
            A[t], B[t], d[t] = self._lin(x[t], u[t])
            x[t + 1] = A[t] @ x[t] + B[t] @ u[t] + d[t]
        zN = self._z(x[-1])
        dz = zN - self.z_target[-1]
        cost += .5 * dz @ self.Qf @ dz
        return x, u, cost, A, B, d

    def update_regularization(self, increase):
        """ilqr.py:198-217 (the decrease branch assigns the misspelt `dhro`, so drho is never
        lowered -- reproduced)."""
        p = self.p
        if increase:
            self.drho = max(self.drho * p.rho_scaling, p.rho_scaling)
            self.rho = max(self.rho * self.drho, p.rho_min)
            if self.rho > p.rho_max:
                self.rho = p.rho_max
        else:
            dhro = min(self.drho / p.rho_scaling, 1.0 / p.rho_scaling)
            self.rho = self.rho * dhro
            if self.rho <= p.rho_min:
                self.rho = p.rho_min

    def dlqr_recursion(self, x, u, A, B, d):
        """ilqr.py:219-300."""
        N, n, m = self.N, self.n, self.m
        H = self.H
        while True:
            Q_u = np.zeros((N, m)); Q_uu = np.zeros((N, m, m))
            K = np.zeros((N, m, n)); k = np.zeros((N, m))
            zN = self._z(x[-1])
            p = H.T @ self.Qf @ (zN - self.z_target[-1])
            P = H.T @ self.Qf @ H
            restart = False
            for t in reversed(range(N)):
                u_prev = (self.u_last if t == 0 else u[t - 1]) if self.p.include_input_var_constraint else np.zeros(m)
                z = self._z(x[t])
                c_xx = H.T @ self.Q @ H
                c_x = H.T @ self.Q @ (z - self.z_target[t])
                c_u = self.R @ (u[t] - u_prev)
                c_uu = self.R
                Q_x = c_x + A[t].T @ p
                Q_u[t] = c_u + B[t].T @ p
                Q_xx = c_xx + A[t].T @ P @ A[t]
                Q_uu[t] = c_uu + B[t].T @ P @ B[t]
                Q_ux = B[t].T @ P @ A[t]
                if self.p.regularize and self.p.state_regularization:
                    Preg = P + self.rho * np.eye(n)
                    Q_uu_t = c_uu + B[t].T @ Preg @ B[t]
                    Q_ux_t = B[t].T @ Preg @ A[t]
                elif self.p.regularize:                                                # ilqr.py:268-270
                    Q_uu_t = Q_uu[t] + self.rho * np.eye(m)
                    Q_ux_t = Q_ux
                else:                                                                  # ilqr.py:272-274
                    Q_uu_t = Q_uu[t]
                    Q_ux_t = Q_ux
                try:
                    np.linalg.cholesky(Q_uu_t)
                except np.linalg.LinAlgError:
                    if self.p.regularize:
                        self.update_regularization(increase=True)
                        restart = True
                        break
                    # (the reference goes on with the inverse of the indefinite matrix, ilqr.py:283-289; so does this statement)
                inv = np.linalg.inv(Q_uu_t)
                K[t] = -inv @ Q_ux_t
                k[t] = -inv @ Q_u[t]
                p = Q_x + K[t].T @ Q_uu[t] @ k[t] + K[t].T @ Q_u[t] + Q_ux.T @ k[t]
                P = Q_xx + K[t].T @ Q_uu[t] @ K[t] + K[t].T @ Q_ux + Q_ux.T @ K[t]
            if restart:
                continue
            self.update_regularization(increase=False)
            break
        return K, k, Q_u, Q_uu

    def solve(self, x0, z_target, u_warm=None):
        """ilqr.py:27-107.  Returns x, u, K and appends (iter, cost, alpha) to self.trace."""
        p = self.p
        self.z_target = z_target
        self.rho, self.drho = p.rho0, p.drho0
        failed_counter = 0
        x_prev = np.zeros((self.N + 1, self.n)); x_prev[0] = x0
        if u_warm is None:
            u_warm = np.zeros((self.N, self.m))
        x, u, cost, A, B, d = self.forward_pass(x_prev, u_warm)
        self.trace = [(0, cost, 1.0)]
        converged, it = False, 0
        K = None
        while not converged and it <= p.max_iter:
            K, k, Q_u, Q_uu = self.dlqr_recursion(x, u, A, B, d)
            prev_cost = cost
            alpha = p.alpha0
            improved = failed = False
            while not improved and not failed:
                improved = True
                xt, ut, ct, At, Bt, dt_ = self.forward_pass(x, u, alpha=alpha, K=K, k=k)
                dcost = 0.
                for t in range(self.N):
                    dcost += alpha * k[t] @ Q_u[t] + alpha ** 2 * .5 * k[t] @ Q_uu[t] @ k[t]
                ratio = (ct - prev_cost) / dcost
                if p.do_linesearch and (ratio <= p.improv_lb or ratio > p.improv_ub):
                    alpha = p.alpha_scaling * alpha
                    improved = False
                    if alpha < p.alpha_min:
                        self.update_regularization(increase=True)
                        self.rho += p.rho_increase_fp
                        failed = True
            if not failed:
                x, u, cost, A, B, d = xt, ut, ct, At, Bt, dt_
                converged = (prev_cost - cost) < p.epsilon and (prev_cost - cost) >= 0
                failed_counter = 0
            else:
                failed_counter += 1
                if failed_counter >= p.counter_limit:
                    converged = True
            it += 1
            self.trace.append((it, cost, alpha))
        return x, u, K


class ILQRGeneric(ILQR):
    """The same loop for any model exposing get_jacobians / x_to_zfyf / H (e.g. the SSM model,
    sofacontrol/SSM/ssm.py): lin_fn(x, u) -> (A_d, B_d, d_d), out_fn(x) -> z, constant H for the cost Jacobians."""

    def __init__(self, lin_fn, out_fn, H, n, m, Q, R, Qf, N):
        self.lin_fn, self.out_fn = lin_fn, out_fn
        self.H, self.Q, self.R, self.Qf, self.N = H, Q, R, Qf, N
        self.n, self.m = n, m
        self.p = ILQRParams()
        self.u_last = np.zeros(m)
        self.z_target = None
        self.trace = []

    def _z(self, x):
        return self.out_fn(x)

    def _lin(self, x, u):
        return self.lin_fn(x, u)
