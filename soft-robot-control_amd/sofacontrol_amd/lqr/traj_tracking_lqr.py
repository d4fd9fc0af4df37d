"""Finite-horizon time-varying LQR (sofacontrol/lqr/traj_tracking_lqr.py:6-48) -- the backward Riccati
recursion runs in one kernel with the TPWL linearisation gathered on the device."""
import ctypes as C

import numpy as np
from scipy.interpolate import interp1d

from .. import _lib


class TrajTrackingLQR:
    def __init__(self, dt, model, cost_params):
        self.dt = dt
        self.model = model
        self.cost_params = cost_params
        self.x_bar = None
        self.u_bar = None

    def compute_policy(self, target):
        K, _ = self.perform_dlqr_recursion(target)
        return self.x_bar, self.u_bar, K

    def perform_dlqr_recursion(self, target):
        x_nom_interp = interp1d(target.t, target.x, axis=0)
        u_nom_interp = interp1d(target.t, target.u, axis=0)
        final_time = target.t[-1]
        nbr_steps = int(final_time / self.dt)
        t_steps = np.arange(nbr_steps) * self.dt
        self.x_bar = x_nom_interp(t_steps)
        self.u_bar = u_nom_interp(t_steps)
        n, m = self.model.get_state_dim(), self.model.get_input_dim()
        K = np.empty((nbr_steps, m, n)); P = np.empty((nbr_steps + 1, n, n))
        xb = _lib.f64(self.x_bar)
        _lib.check(_lib.lib().sric_tvlqr_tpwl(self.model.handle_for(self.dt), _lib.dptr(xb), C.c_int(nbr_steps),
                                              _lib.dptr(_lib.f64(self.cost_params.Q)), _lib.dptr(_lib.f64(self.cost_params.R)),
                                              _lib.dptr(K), _lib.dptr(P)), 'sric_tvlqr_tpwl')
        return K, P
