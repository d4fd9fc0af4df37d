"""Infinite-horizon discrete LQR gains on the device (sofacontrol/lqr/lqr.py:6-54)."""
import ctypes as C

import numpy as np

from .. import _lib


def _fixed_point(A, B, Q, R, tol, max_iter):
    A = _lib.f64(np.atleast_3d(A).reshape(-1, A.shape[-2], A.shape[-1]))
    B = _lib.f64(np.atleast_3d(B).reshape(-1, B.shape[-2], B.shape[-1]))
    batch, n, m = B.shape
    L = np.empty((batch, m, n)); P = np.empty((batch, n, n)); it = np.empty(batch, dtype=np.int32)
    _lib.check(_lib.lib().sric_dare_fixed_point(_lib.dptr(A), _lib.dptr(B), C.c_int64(batch), C.c_int(n), C.c_int(m),
                                                _lib.dptr(_lib.f64(Q)), _lib.dptr(_lib.f64(R)), C.c_double(tol),
                                                C.c_int(max_iter), _lib.dptr(L), _lib.dptr(P), _lib.iptr(it)),
               'sric_dare_fixed_point')
    return L, P, it


def solve_riccati(A, B, Q, R):
    """lqr.py:6-21: fixed-point DARE until ||L - L_old||_F <= 1e-4; returns (L, P), u = +L x."""
    L, P, _ = _fixed_point(A, B, Q, R, 1e-4, 1000000)
    return L[0], P[0]


def _doubling(A, B, Q, R, tol, max_iter):
    A = _lib.f64(np.atleast_3d(A).reshape(-1, A.shape[-2], A.shape[-1]))
    B = _lib.f64(np.atleast_3d(B).reshape(-1, B.shape[-2], B.shape[-1]))
    batch, n, m = B.shape
    L = np.empty((batch, m, n)); P = np.empty((batch, n, n)); it = np.empty(batch, dtype=np.int32)
    _lib.check(_lib.lib().sric_dare(_lib.dptr(A), _lib.dptr(B), C.c_int64(batch), C.c_int(n), C.c_int(m),
                                    _lib.dptr(_lib.f64(Q)), _lib.dptr(_lib.f64(R)), C.c_double(tol), C.c_int(max_iter),
                                    _lib.dptr(L), _lib.dptr(P), _lib.iptr(it)), 'sric_dare')
    return L, P, it


def dare(Ad, Bd, Q, R):
    """lqr.py:24-31 (scipy.linalg.solve_discrete_are in the reference): the stabilising DARE solution and its gain
    K = -(R + B'PB)^-1 B'PA, by the structure-preserving doubling algorithm on the device (`sric_dare`)."""
    L, P, _ = _doubling(Ad, Bd, Q, R, 1e-14, 100)
    return L[0], P[0]


def dare_batch(Ad, Bd, Q, R, tol=1e-14):
    """Gains for a stack of (A_d, B_d) pairs in one launch (the per-point gains of the scp controller,
    tpwl/controllers.py:238-246)."""
    L, P, _ = _doubling(Ad, Bd, Q, R, tol, 100)
    return L, P


class DLQR:
    """lqr.py:34-54."""

    def __init__(self, dt, model, cost_params):
        self.dt = dt
        self.model = model
        self.cost_params = cost_params

    def compute_policy(self, target):
        u_nom = np.atleast_1d(target.u)
        x_nom = target.x
        K = self.compute_gain_matrix(target.A, target.B, self.cost_params.Q, self.cost_params.R)
        return x_nom, u_nom, K

    def compute_gain_matrix(self, A, B, Q, R):
        Ad, Bd, _ = self.model.discretize_dynamics(A_c=A, B_c=B, d_c=np.zeros(self.model.get_state_dim()), dt=self.dt)
        K, _ = solve_riccati(Ad, Bd, Q, R)
        return K
