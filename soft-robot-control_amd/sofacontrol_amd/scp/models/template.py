"""Model protocol consumed by GuSTO (sofacontrol/scp/models/template.py:1-88)."""
import numpy as np


class TemplateModel:
    def __init__(self):
        self.H = None
        self.n_x = None
        self.n_u = None
        self.n_z = None
        self.nonlinear_observer = False

    def get_continuous_dynamics(self, x, u):
        raise RuntimeError('Must be subclassed and implemented')

    def get_discrete_dynamics(self, x, u, dt):
        raise RuntimeError('Must be subclassed and implemented')

    def get_characteristic_vals(self):
        return np.ones(self.n_x), np.ones(self.n_x)

    def rollout(self, x0, u, dt):
        raise RuntimeError('Must be subclassed and implemented')
