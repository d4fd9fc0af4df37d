"""Planar unicycle, the reference's stand-alone GuSTO demonstration model (sofacontrol/scp/models/dubins_car.py,
driven by sofacontrol/scp/example.py): state (p_x, p_y, heading), inputs (speed, turn rate), every state a
performance output.  A generic model for GuSTO's host loop: the linearisations are formed here per stage, the horizon QP
runs on the device."""
import numpy as np

from .template import TemplateModel


def _heading_frame(x, u):
    """Unit heading vector, its derivative w.r.t. the heading, and the vector field of the unicycle."""
    c, s = np.cos(x[2]), np.sin(x[2])
    return c, s, np.array([u[0] * c, u[0] * s, u[1]])


class DubinsCar(TemplateModel):
    def __init__(self):
        super().__init__()
        self.n_x, self.n_u, self.n_z = 3, 2, 3
        self.H = np.eye(3)

    def get_continuous_dynamics(self, x, u):
        """f(x, u), df/dx, df/du (dubins_car.py:16-31)."""
        c, s, f = _heading_frame(x, u)
        A = np.zeros((3, 3))
        A[:2, 2] = u[0] * np.array([-s, c])          # only the heading moves the velocity direction
        B = np.array([[c, 0.0], [s, 0.0], [0.0, 1.0]])
        return f, A, B

    def get_discrete_dynamics(self, x, u, dt):
        """Forward-Euler step of the first-order expansion about (x, u): x+ = A_d x + B_d u + d_d (dubins_car.py:33-42)."""
        f, A, B = self.get_continuous_dynamics(x, u)
        return np.eye(3) + dt * A, dt * B, dt * (f - A @ x - B @ u)

    def get_next_state(self, x, u, dt):
        return x + dt * _heading_frame(x, u)[2]

    def rollout(self, x0, u, dt):
        """Forward-Euler simulation of the nonlinear model (dubins_car.py:51-65): states only, (N + 1, 3)."""
        u = np.asarray(u, dtype=np.float64)
        x = np.empty((u.shape[0] + 1, 3))
        x[0] = x0
        for k in range(u.shape[0]):
            x[k + 1] = self.get_next_state(x[k], u[k], dt)
        return x
