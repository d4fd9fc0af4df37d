"""Observers (sofacontrol/tpwl/observer.py).  FullStateObserver (lines 3-30) is pure bookkeeping.  The
DiscreteEKFObserver (lines 33-126) is next-tier and not provided yet."""


class FullStateObserver:
    def __init__(self, n_x, H=None):
        self.x = None
        self.z = None
        self.meas_dim = n_x
        self.state_dim = n_x
        self.H = H

    def get_meas_dim(self):
        return self.meas_dim

    def get_observer_params(self):
        return {'meas_dim': self.meas_dim, 'state_dim': self.state_dim}

    def update(self, u, y, dt, x=None):
        self.x = x
        self.z = self.H @ x if self.H is not None else x


class DiscreteEKFObserver:
    def __init__(self, *a, **k):
        raise NotImplementedError('DiscreteEKFObserver is next-tier (fused with the projection step); use FullStateObserver')
