"""GuSTO adapter of the SSM model (sofacontrol/scp/models/ssm.py:7-93)."""
import numpy as np

from .template import TemplateModel


class SSMGuSTO(TemplateModel):
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
        self.nonlinear_observer = self.dyn_sys.nonlinear_observer

    def get_continuous_dynamics(self, x, u):
        """models/ssm.py:35-54: f = A x + B u + d with the continuous Jacobians at (x, u)."""
        A, B, d = self.dyn_sys.get_continuous_jacobians(x, u=u)
        f = A @ x + B @ u + d
        return f, A, B

    def get_discrete_dynamics(self, x, u, dt):
        return self.dyn_sys.get_jacobians(x, dt=dt, u=u)

    def get_observer_jacobians(self, x, u, dt):
        return self.dyn_sys.get_observer_jacobians(x)

    # ---- batched forms used by the GuSTO host loop: one device call for the whole horizon
    def get_discrete_dynamics_batch(self, X, U, dt):
        return self.dyn_sys.get_jacobians(np.asarray(X), dt=dt, u=np.asarray(U))

    def get_continuous_dynamics_batch(self, X, U):
        X, U = np.asarray(X), np.asarray(U)
        A, B, d = self.dyn_sys.get_continuous_jacobians(X, u=U)
        f = np.einsum('kij,kj->ki', A, X) + np.einsum('kij,kj->ki', B, U) + d
        return f, A, B

    def get_observer_jacobians_batch(self, X, dt):
        return self.dyn_sys.get_observer_jacobians(np.asarray(X))

    def get_characteristic_vals(self):
        return np.ones(self.n_x), np.ones(self.n_x)

    def rollout(self, x0, u, dt):
        return self.dyn_sys.rollout(x0, u, dt)
