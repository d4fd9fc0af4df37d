"""Observers (sofacontrol/tpwl/observer.py).  FullStateObserver (lines 3-30) is pure bookkeeping; the
DiscreteEKFObserver (lines 33-126) keeps its estimate and covariance in HBM and runs one kernel per step
(`sekf_step`, csrc/observer.hip)."""
import ctypes as C

import numpy as np

from .. import _lib


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
    """observer.py:33-126.  `Sigma` is read back from the device on access."""

    def __init__(self, dyn_sys, **kwargs):
        self.dyn_sys = dyn_sys
        if self.dyn_sys.C is None:
            raise RuntimeError('Need to set meas. model in dyn_sys')
        self.C = self.dyn_sys.C
        self.state_dim = self.dyn_sys.get_state_dim()
        self.meas_dim = self.C.shape[0]
        self._Sigma0 = np.array(kwargs.get('Sigma0', np.eye(self.state_dim)), dtype=np.float64)
        self.W = kwargs.get('W', 100 * np.eye(self.state_dim))
        self.V = kwargs.get('V', np.eye(self.meas_dim))
        self._h = C.c_void_p()
        self._filter_dt = None
        self.x = None
        self._z = None
        self._make_filter(None, self._Sigma0, None)
        self.initialize(self.dyn_sys.rom.x_ref)

    def _make_filter(self, dt, Sigma, x):
        """(Re)create the device filter on the model handle whose tables are discretised at `dt` (the planner and
        the simulation usually run different time steps: each keeps its own tables, tpwl.handle_for)."""
        if self._h:
            _lib.lib().sekf_destroy(self._h)
            self._h = C.c_void_p()
        nn = getattr(self.dyn_sys, 'tpwl_method', 'nn') == 'nn'
        mh = self.dyn_sys.handle_for(dt if nn else None)
        Cm, yr = _lib.f64(self.C), _lib.f64(self.dyn_sys.y_ref)
        S0, W, V = _lib.f64(Sigma), _lib.f64(self.W), _lib.f64(self.V)
        _lib.check(_lib.lib().sekf_create(C.byref(self._h), mh, _lib.dptr(Cm), _lib.dptr(yr),
                                          C.c_int(self.meas_dim), _lib.dptr(S0), _lib.dptr(W), _lib.dptr(V)),
                   'sekf_create')
        if x is not None:
            xx = _lib.f64(x)
            _lib.check(_lib.lib().sekf_set_state(self._h, _lib.dptr(xx), None), 'sekf_set_state')
        self._filter_dt = dt

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            try:
                _lib.lib().sekf_destroy(h)
            except Exception:
                pass
            self._h = None

    def get_meas_dim(self):
        return self.meas_dim

    def get_observer_params(self):
        return {'W': self.W, 'V': self.V, 'meas_dim': self.meas_dim, 'state_dim': self.state_dim,
                'C': self.C, 'H': self.dyn_sys.H}

    @property
    def Sigma(self):
        S = np.empty((self.state_dim, self.state_dim))
        _lib.check(_lib.lib().sekf_get_state(self._h, None, _lib.dptr(S)), 'sekf_get_state')
        return S

    @Sigma.setter
    def Sigma(self, S):
        S = _lib.f64(S)
        _lib.check(_lib.lib().sekf_set_state(self._h, None, _lib.dptr(S)), 'sekf_set_state')

    # z follows x (observer.py:92-95, 123-126); it is evaluated when read, so a control loop that only uses the
    # estimate does not pay a sparse product per simulation step
    @property
    def z(self):
        if self._z is None and self.x is not None:
            if self.dyn_sys.H is not None:
                self._z = self.dyn_sys.x_to_zfyf(self.x, zf=True)
            else:
                self._z = self.dyn_sys.x_to_zfyf(self.x, yf=True)
        return self._z

    @z.setter
    def z(self, value):
        self._z = value

    def _set_z(self):
        self._z = None

    def initialize(self, xf):
        """observer.py:76-86."""
        self.x = self.dyn_sys.rom.compute_RO_state(xf=xf)
        x = _lib.f64(self.x)
        _lib.check(_lib.lib().sekf_set_state(self._h, _lib.dptr(x), None), 'sekf_set_state')
        self._set_z()

    def _step(self, u, y, dt):
        A = B = d = None
        if u is not None:
            if getattr(self.dyn_sys, 'tpwl_method', 'nn') == 'nn':
                if self._filter_dt != dt:                   # first predictor step, or a new time step
                    self._make_filter(dt, self.Sigma, self.x)
            else:
                A, B, d = [_lib.f64(a) for a in self.dyn_sys.get_jacobians(self.x, dt)]
            u = _lib.f64(u)
        if y is not None:
            y = _lib.f64(y)
        x = np.empty(self.state_dim)
        _lib.check(_lib.lib().sekf_step(self._h, _lib.dptr(u), _lib.dptr(y), _lib.dptr(A), _lib.dptr(B),
                                        _lib.dptr(d), _lib.dptr(x)), 'sekf_step')
        self.x = x

    def update_projected(self, rom, xf, u, y, dt):
        """The per-simulation-step pair rom.compute_RO_state(xf=xf); update(u, y, dt) in one library call
        (`sekf_step_projected`: the projection runs on a side stream beside the filter kernel).  Returns the projected
        state.  Models that hand the filter external Jacobians (weighting modes) take the two separate calls."""
        if u is None or getattr(self.dyn_sys, 'tpwl_method', 'nn') != 'nn' or getattr(rom, 'handle', None) is None:
            x_reduced = rom.compute_RO_state(xf=xf)
            self.update(u, y, dt)
            return x_reduced
        if self._filter_dt != dt:
            self._make_filter(dt, self.Sigma, self.x)
        xf, u, y = _lib.f64(xf), _lib.f64(u), _lib.f64(y)
        n_f, r = rom.U.shape
        if xf.shape != (2 * n_f,):
            raise RuntimeError('sekf_step_projected: expected a full-order state of %d entries, got %s'
                               % (2 * n_f, xf.shape))
        x_reduced, x = np.empty(2 * r), np.empty(self.state_dim)
        _lib.check(_lib.lib().sekf_step_projected(self._h, rom.handle, _lib.dptr(xf), _lib.dptr(u), _lib.dptr(y),
                                                  _lib.dptr(x_reduced), _lib.dptr(x)), 'sekf_step_projected')
        self.x = x
        self._set_z()
        return x_reduced

    def update(self, u, y, dt, **kwargs):
        """observer.py:88-95: predictor + filter update in one kernel."""
        self._step(u, y, dt)
        self._set_z()

    def predict_state(self, u, dt):
        """observer.py:97-106."""
        self._step(u, None, dt)

    def update_state(self, y):
        """observer.py:108-126."""
        self._step(None, y, None)
        self._set_z()
        return self.x
