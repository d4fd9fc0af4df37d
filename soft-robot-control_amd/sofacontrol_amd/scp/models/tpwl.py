"""GuSTO adapter of the TPWL model (sofacontrol/scp/models/tpwl.py:7-96)."""
import ctypes as C

import numpy as np

from ... import _lib
from .template import TemplateModel


class TPWLGuSTO(TemplateModel):
    def __init__(self, dyn_sys):
        super().__init__()
        self.dyn_sys = dyn_sys
        if self.dyn_sys.H is not None:
            self.H = self.dyn_sys.H
        else:
            raise RuntimeError('dyn_sys must have output model specified')
        self.n_x = self.dyn_sys.get_state_dim()
        self.n_u = self.dyn_sys.get_input_dim()
        self.n_z = self.H.shape[0]
        self.nonlinear_observer = False

    def get_continuous_dynamics(self, x, u):
        """models/tpwl.py:32-50: f = A_i x + B_i u + d_i at the nearest stored point."""
        A, B, d = self.dyn_sys.get_jacobians(x)
        return A @ x + B @ u + d, A, B

    def get_discrete_dynamics(self, x, u, dt):
        return self.dyn_sys.get_jacobians(x, dt=dt)

    # ---- batched forms used by the GuSTO host loop (weighting-mode models): one device call per horizon
    def get_discrete_dynamics_batch(self, X, U, dt):
        A, B, d, _ = self.dyn_sys.linearize_batch(np.asarray(X), dt)
        return A, B, d

    def get_continuous_dynamics_batch(self, X, U):
        X, U = np.asarray(X), np.asarray(U)
        A, B, d, _ = self.dyn_sys.linearize_batch(X)
        f = np.einsum('kij,kj->ki', A, X) + np.einsum('kij,kj->ki', B, U) + d
        return f, A, B

    def pre_discretize(self, dt):
        self.dyn_sys.pre_discretize(dt)

    def get_characteristic_vals(self):
        """models/tpwl.py:66-84 (one kernel over the stored points)."""
        n = self.n_x
        if self.dyn_sys.tpwl_method != 'nn':
            from ... import utils as scutils
            tabs = self.dyn_sys._tabs
            x = scutils.qv2x(tabs[0], tabs[1])
            A, B, d, _ = self.dyn_sys.linearize_batch(x)
            f = np.einsum('bij,bj->bi', A, x) + np.einsum('bij,bj->bi', B, tabs[2]) + d
            return np.abs(x).max(axis=0), np.abs(f).max(axis=0)
        xc, fc = np.empty(n), np.empty(n)
        _lib.check(_lib.lib().stpwl_characteristic(self.dyn_sys.handle_for(None), _lib.dptr(xc), _lib.dptr(fc)),
                   'stpwl_characteristic')
        return xc, fc

    def rollout(self, x0, u, dt):
        return self.dyn_sys.rollout(x0, u, dt)
