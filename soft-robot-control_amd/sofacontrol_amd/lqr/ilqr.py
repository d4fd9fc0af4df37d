"""iLQR on the device (sofacontrol/lqr/ilqr.py:6-300): the whole `ilqr_computation` -- forward passes with
nearest-point TPWL dynamics, backward Riccati passes with the reference's regularisation schedule, line
search -- is one persistent kernel (csrc/lqr.hip), one workgroup per problem."""
import ctypes as C

import numpy as np

from .. import _lib
from .config import iLQRConfig


class iLQR:
    def __init__(self, dt, model, cost_params, planning_horizon, **kwargs):
        self.params = iLQRConfig()
        self.dt = dt
        self.model = model
        self.planning_horizon = planning_horizon
        self.cost_params = cost_params
        self.state_dim = model.get_state_dim()
        self.input_dim = model.get_input_dim()
        self.z_target = None
        self.u_last = np.zeros(self.input_dim)
        self.cost = None
        self.iters = None

    def set_target(self, z_target):
        self.z_target = z_target.copy()

    def set_u_last(self, u_last):
        self.u_last = u_last.copy()

    def _params(self):
        p = self.params
        # the four switches of config.py:6-9, 31 are kernel parameters (csrc/lqr.hip); regularize = False with a Q_uu that is
        # not positive definite ends the solve with iters = -1 (the reference goes on with the inverse of an indefinite matrix)
        return _lib.SIlqrParams(p.max_iter, p.epsilon, p.alpha0, p.alpha_scaling, p.improv_lb, float(p.improv_ub),
                                p.alpha_min, p.counter_limit, p.rho0, p.drho0, p.rho_scaling, p.rho_increase_fp,
                                p.rho_max, p.rho_min, int(bool(p.include_input_var_constraint)), int(bool(p.do_linesearch)),
                                int(bool(p.regularize)), int(bool(p.state_regularization)))

    def ilqr_computation(self, x0, u_warmstart=None):
        """ilqr.py:27-107; batched when x0 is (B, n_x) (z_target (B, N+1, n_z), u_warmstart (B, N, n_u))."""
        from ..SSM.ssm import SSM
        is_ssm = isinstance(self.model, SSM)
        N, n, m = self.planning_horizon, self.state_dim, self.input_dim
        x0a = _lib.f64(np.atleast_2d(x0))
        Bn = x0a.shape[0]
        single = np.ndim(x0) == 1
        zt = _lib.f64(np.broadcast_to(self.z_target, (Bn,) + self.z_target.shape[-2:]))
        uw = None if u_warmstart is None else _lib.f64(np.broadcast_to(u_warmstart, (Bn, N, m)))
        ul = _lib.f64(np.broadcast_to(self.u_last, (Bn, m)))
        x = np.empty((Bn, N + 1, n)); u = np.empty((Bn, N, m)); K = np.empty((Bn, N, m, n))
        cost = np.empty(Bn); iters = np.empty(Bn, dtype=np.int32)
        par = self._params()
        cp = self.cost_params
        if is_ssm:
            # the model's bookkeeping H (zeros unless the user set it, ssm.py:69-70) enters the cost Jacobians
            _lib.check(_lib.lib().sssm_set_output(self.model.handle, _lib.dptr(_lib.f64(self.model.H))), 'sssm_set_output')
            _lib.check(_lib.lib().silqr_solve_ssm(self.model.handle, C.c_int(self.model._mode()), C.c_double(self.dt),
                                                  C.c_int(N), C.c_int64(Bn), _lib.dptr(x0a), _lib.dptr(zt),
                                                  _lib.dptr(uw), _lib.dptr(ul), _lib.dptr(_lib.f64(cp.Q)),
                                                  _lib.dptr(_lib.f64(cp.R)), _lib.dptr(_lib.f64(cp.Qf)), C.byref(par),
                                                  _lib.dptr(x), _lib.dptr(u), _lib.dptr(K), _lib.dptr(cost),
                                                  _lib.iptr(iters)), 'silqr_solve_ssm')
            self.cost, self.iters = cost, iters
            return (x[0], u[0], K[0]) if single else (x, u, K)
        _lib.check(_lib.lib().silqr_solve(self.model.handle_for(self.dt), C.c_int(N), C.c_int64(Bn), _lib.dptr(x0a), _lib.dptr(zt),
                                          _lib.dptr(uw), _lib.dptr(ul), _lib.dptr(_lib.f64(cp.Q)), _lib.dptr(_lib.f64(cp.R)),
                                          _lib.dptr(_lib.f64(cp.Qf)), C.byref(par), _lib.dptr(x), _lib.dptr(u),
                                          _lib.dptr(K), _lib.dptr(cost), _lib.iptr(iters)), 'silqr_solve')
        self.cost, self.iters = cost, iters
        if single:
            return x[0], u[0], K[0]
        return x, u, K
