"""TPWL piecewise-affine reduced model on MI355X -- surface of sofacontrol/tpwl/tpwl.py (TPWL,
TPWLATV).  The point tables live in HBM; nearest-point search, table gather and rollouts are HIP
kernels (csrc/tpwl.hip).  Discretisation -- of the P stored points (pre_discretize, one-off per dt, tpwl.py:299-322) and of the
blended model that weighting mode re-discretises at every state (tpwl.py:244-250) -- is a batched kernel too (csrc/discretize.hip:
Gauss-Jordan for be / bil, scaling and squaring with the [13/13] Pade approximant for zoh)."""
import ctypes as C

import numpy as np

from .. import _lib
from .. import utils as scutils
from ..mor import pod

DISCR_METHOD = 'zoh'
TPWL_METHOD = 'nn'
DISCR_DICT = {'fe': 'forward Euler', 'be': 'implicit Euler', 'bil': 'bilinear transform', 'zoh': 'zero-order hold'}


class TPWL:
    """sofacontrol/tpwl/tpwl.py:14-216."""

    def __init__(self, data, params=None, Cf=None, Hf=None, **kwargs):
        self.tpwl_dict = data if isinstance(data, dict) else scutils.load_data(data)
        self.num_points = len(self.tpwl_dict['q'])
        self.discr_method = kwargs.get('discr_method', 'fe')
        if self.tpwl_dict['rom_info']['type'] == 'POD':
            self.rom = pod.POD(self.tpwl_dict['rom_info'])
        else:
            raise NotImplementedError("Unknown ROM type")
        q = np.asarray(self.tpwl_dict['q'], dtype=np.float64)
        u = np.asarray(self.tpwl_dict['u'], dtype=np.float64)
        self.state_dim = q.shape[-1] * 2
        self.input_dim = u.shape[-1]
        if params is None:
            params = dict()
        self.tpwl_method = params.get('tpwl_method', TPWL_METHOD)
        self.beta_weighting = params.get('beta_weighting', None)
        self.dist_weights = params.get('dist_weights')
        if self.dist_weights is None:
            raise RuntimeError("params['dist_weights'] = {'q': .., 'v': ..} is required (tpwl.py:165-166)")

        self._h = C.c_void_p()
        self._dt_handles = {}          # dt -> device handle with the tables discretised at dt (never mutated)
        f = _lib.f64
        self._tabs = [f(q), f(self.tpwl_dict['v']), f(u), f(self.tpwl_dict['A_c']), f(self.tpwl_dict['B_c']),
                      f(self.tpwl_dict['d_c'])]
        _lib.check(_lib.lib().stpwl_create(
            C.byref(self._h), C.c_int(self.num_points), C.c_int(q.shape[-1]), C.c_int(self.input_dim),
            *[_lib.dptr(t) for t in self._tabs], None, None, None,
            C.c_double(float(self.dist_weights['q'])), C.c_double(float(self.dist_weights['v']))), 'stpwl_create')

        if Cf is not None:
            self.set_measurement_model(Cf)
        else:
            self.C = None
            self.y_ref = None
            self.meas_dim = None
        if Hf is not None:
            self.set_output_model(Hf)
        else:
            self.H = None
            self.z_ref = None
            self.output_dim = None
        self.nonlinear_observer = False
        self.pre_discretized_dt = None
        self.A_d = None
        self.B_d = None
        self.d_d = None

    def __del__(self):
        try:
            for h in list(getattr(self, '_dt_handles', {}).values()) + [self._h]:
                if h:
                    _lib.lib().stpwl_destroy(h)
            self._dt_handles = {}
            self._h = C.c_void_p()
        except Exception:
            pass

    @property
    def handle(self):
        """Device handle: the one pre-discretised with `pre_discretized_dt` if there is one, else the continuous
        one."""
        if self.pre_discretized_dt is not None:
            return self.handle_for(self.pre_discretized_dt)
        return self._h

    def handle_for(self, dt, tables=None):
        """Device handle whose discrete tables are discretised at `dt` (created on first use, then immutable: a
        GuSTO plan, an observer and a rollout with different time steps each keep their own tables -- the
        reference discretises per call in that case, tpwl.py:260-265).  dt = None: the continuous handle."""
        if dt is None:
            return self._h
        key = float(dt)
        h = self._dt_handles.get(key)
        if h is not None:
            return h
        if self.tpwl_method != 'nn':
            raise RuntimeError('tpwl method should be nn to pre-discretize')
        if tables is None:
            Ac, Bc, dc = self._tabs[3], self._tabs[4], self._tabs[5]
            A_d, B_d, d_d = self.discretize_batch(Ac, Bc, dc, dt)
        else:
            A_d, B_d, d_d = tables
        Ad, Bd, dd = _lib.f64(np.stack(A_d)), _lib.f64(np.stack(B_d)), _lib.f64(np.stack(d_d))
        h = C.c_void_p()
        _lib.check(_lib.lib().stpwl_create(
            C.byref(h), C.c_int(self.num_points), C.c_int(self._tabs[0].shape[-1]), C.c_int(self.input_dim),
            *[_lib.dptr(t) for t in self._tabs], _lib.dptr(Ad), _lib.dptr(Bd), _lib.dptr(dd),
            C.c_double(float(self.dist_weights['q'])), C.c_double(float(self.dist_weights['v']))), 'stpwl_create')
        if getattr(self, 'H', None) is not None:
            _lib.check(_lib.lib().stpwl_set_output(h, _lib.dptr(self.H), _lib.dptr(self.z_ref), C.c_int(self.output_dim)),
                       'stpwl_set_output')
        self._dt_handles[key] = h
        return h

    def update_state(self, x, u, dt):
        raise NotImplementedError("update_state must be overriden by a child class")

    def get_jacobians(self, x, dt=None):
        raise NotImplementedError("get_jacobians must be overriden by a child class")

    def set_measurement_model(self, Cf):
        """tpwl.py:81-84."""
        self.C = np.asarray(Cf @ self.rom.V)
        self.y_ref = np.asarray(Cf @ self.rom.x_ref).ravel()
        self.meas_dim = self.C.shape[0]

    def set_output_model(self, Hf):
        """tpwl.py:86-89: H = Hf V, z_ref = Hf x_ref (selector rows: a gather, done on the host once)."""
        self.H = np.ascontiguousarray(np.asarray(Hf @ self.rom.V), dtype=np.float64)
        self.z_ref = np.ascontiguousarray(np.asarray(Hf @ self.rom.x_ref).ravel(), dtype=np.float64)
        self.output_dim = self.H.shape[0]
        for h in [self._h] + list(self._dt_handles.values()):
            _lib.check(_lib.lib().stpwl_set_output(h, _lib.dptr(self.H), _lib.dptr(self.z_ref),
                                                   C.c_int(self.output_dim)), 'stpwl_set_output')

    def zfyf_to_zy(self, zf=None, yf=None):
        if zf is not None and self.z_ref is not None:
            return zf - self.z_ref
        elif yf is not None and self.y_ref is not None:
            return yf - self.y_ref
        raise RuntimeError('Need to set output or meas. model')

    def zy_to_zfyf(self, z=None, y=None):
        if z is not None and self.z_ref is not None:
            return z + self.z_ref
        elif y is not None and self.y_ref is not None:
            return y + self.y_ref
        raise RuntimeError('Need to set output or meas. model')

    def x_to_zfyf(self, x, zf=False, yf=False):
        if zf and self.H is not None:
            return np.transpose(self.H @ x.T) + self.z_ref
        elif yf and self.C is not None:
            return np.transpose(self.C @ x.T) + self.y_ref
        raise RuntimeError('Need to set output or meas. model')

    def x_to_zy(self, x, z=False, y=False):
        if z and self.H is not None:
            return np.transpose(self.H @ x.T)
        elif y and self.C is not None:
            return np.transpose(self.C @ x.T)
        raise RuntimeError('Need to set output or meas. model')

    def get_state_dim(self):
        return self.state_dim

    def get_input_dim(self):
        return self.input_dim

    def get_output_dim(self):
        return self.output_dim

    def get_meas_dim(self):
        return self.meas_dim

    def get_rom_info(self):
        return self.tpwl_dict['rom_info']

    def get_sim_params(self):
        return {'beta_weighting': self.beta_weighting, 'discr_method': self.discr_method,
                'tpwl_method': self.tpwl_method, 'dist_weights': self.dist_weights}

    def calc_nearest_point(self, x):
        """tpwl.py:160-168 (one state) -- accepts (B, n_x) for a batch."""
        X = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        idx = np.empty(X.shape[0], dtype=np.int32)
        _lib.check(_lib.lib().stpwl_nearest(self._h, _lib.dptr(X), C.c_int64(X.shape[0]), _lib.iptr(idx)),
                   'stpwl_nearest')
        return int(idx[0]) if np.ndim(x) == 1 else idx

    def calc_weighting_factors(self, x):
        """tpwl.py:170-191 (one state) -- accepts (B, n_x) for a batch -> (B, P)."""
        X = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        W = np.empty((X.shape[0], self.num_points))
        _lib.check(_lib.lib().stpwl_weights(self._h, _lib.dptr(X), C.c_int64(X.shape[0]),
                                            C.c_double(self.beta_weighting), _lib.dptr(W)), 'stpwl_weights')
        return W[0] if np.ndim(x) == 1 else W

    def rollout(self, x0, u, dt):
        """tpwl.py:193-216; x0 (n_x,), u (N, n_u) -> x (N+1, n_x), z (N+1, n_z) (z includes z_ref).
        Batched form: x0 (B, n_x), u (B, N, n_u)."""
        if self.tpwl_method == 'weighting':
            # the blended model is re-discretised at every step (tpwl.py:244-250): step-wise like the reference
            u = np.asarray(u, dtype=np.float64)
            N = u.shape[0]
            x = np.zeros((N + 1, self.state_dim))
            x[0, :] = x0
            for i in range(N):
                x[i + 1, :] = self.update_state(x[i, :], u[i, :], dt)
            z = self.x_to_zfyf(x, zf=True) if self.H is not None else None
            return x, z
        h = self.handle_for(dt)
        x0a = np.ascontiguousarray(np.atleast_2d(x0), dtype=np.float64)
        ua = np.ascontiguousarray(u, dtype=np.float64)
        single = ua.ndim == 2
        if single:
            ua = ua[None]
        Bn, N = ua.shape[0], ua.shape[1]
        X = np.empty((Bn, N + 1, self.state_dim))
        Z = np.empty((Bn, N + 1, self.output_dim)) if self.H is not None else None
        _lib.check(_lib.lib().stpwl_rollout(h, _lib.dptr(x0a), _lib.dptr(ua), C.c_int(N), C.c_int64(Bn),
                                            _lib.dptr(X), _lib.dptr(Z)), 'stpwl_rollout')
        if single:
            return X[0], (Z[0] if Z is not None else None)
        return X, Z


class TPWLATV(TPWL):
    """sofacontrol/tpwl/tpwl.py:219-342."""

    def __init__(self, data, params=None, Cf=None, Hf=None, **kwargs):
        super().__init__(data, params, Cf=Cf, Hf=Hf, **kwargs)
        self.ref_point = None
        if self.tpwl_method not in ('nn', 'weighting'):
            raise RuntimeError('tpwl method should be nn or weighting')
        if self.tpwl_method == 'weighting' and self.beta_weighting is None:
            raise RuntimeError("tpwl_method='weighting' needs params['beta_weighting']")

    def update_state(self, x, u, dt):
        A_d, B_d, d_d = self.get_jacobians(x, dt)
        return self.update_dynamics(x, u, A_d, B_d, d_d)

    def get_jacobians(self, x, dt=None, u=None):
        """tpwl.py:236-270.  nn: (A_d, B_d, d_d)[i] if dt is given, else continuous (A_c, B_c, d_c)[i];
        weighting: softmin-blended continuous tables (device), discretised on the host when dt is given."""
        if self.tpwl_method == 'weighting':
            A, B, d, _ = self.linearize_batch(np.atleast_2d(x), dt)
            return A[0], B[0], d[0]
        h = self.handle_for(dt)
        X = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        n, m = self.state_dim, self.input_dim
        A = np.empty((1, n, n)); B = np.empty((1, n, m)); d = np.empty((1, n))
        idx = np.empty(1, dtype=np.int32)
        _lib.check(_lib.lib().stpwl_linearize(h, _lib.dptr(X), C.c_int64(1), C.c_int(dt is not None),
                                              _lib.dptr(A), _lib.dptr(B), _lib.dptr(d), _lib.iptr(idx)),
                   'stpwl_linearize')
        self.ref_point = int(idx[0])
        return A[0], B[0], d[0]

    def linearize_batch(self, X, dt=None):
        """Batched get_jacobians for X (B, n_x) -> A (B,n,n), B (B,n,m), d (B,n), idx (B,) (nn) or the
        weights (B, P) (weighting)."""
        if self.tpwl_method == 'weighting':
            X = np.ascontiguousarray(X, dtype=np.float64)
            Bn, n, m = X.shape[0], self.state_dim, self.input_dim
            A = np.empty((Bn, n, n)); B = np.empty((Bn, n, m)); d = np.empty((Bn, n))
            W = np.empty((Bn, self.num_points))
            _lib.check(_lib.lib().stpwl_linearize_weighted(self._h, _lib.dptr(X), C.c_int64(Bn),
                                                           C.c_double(self.beta_weighting), _lib.dptr(A),
                                                           _lib.dptr(B), _lib.dptr(d), _lib.dptr(W)),
                       'stpwl_linearize_weighted')
            if dt is not None:
                A, B, d = self.discretize_batch(A, B, d, dt)          # the blended models of the whole batch in one launch
            return A, B, d, W
        h = self.handle_for(dt)
        X = np.ascontiguousarray(X, dtype=np.float64)
        Bn, n, m = X.shape[0], self.state_dim, self.input_dim
        A = np.empty((Bn, n, n)); B = np.empty((Bn, n, m)); d = np.empty((Bn, n))
        idx = np.empty(Bn, dtype=np.int32)
        _lib.check(_lib.lib().stpwl_linearize(h, _lib.dptr(X), C.c_int64(Bn), C.c_int(dt is not None),
                                              _lib.dptr(A), _lib.dptr(B), _lib.dptr(d), _lib.iptr(idx)),
                   'stpwl_linearize')
        return A, B, d, idx

    def discretize_batch(self, A_c, B_c, d_c, dt):
        """tpwl.py:272-297 for a stack of models (B, n, n), (B, n, m), (B, n) in one launch of csrc/discretize.hip: fe / be / bil by
        one Gauss-Jordan elimination per model, zoh (utils.py:302-335) by scaling and squaring with the [13/13] Pade approximant."""
        methods = {'fe': 0, 'be': 1, 'bil': 2, 'zoh': 3}
        if self.discr_method not in methods:
            raise RuntimeError('self.discr_method must be in [fe, be, bil, zoh]')
        A = _lib.f64(np.asarray(A_c)); Bm = _lib.f64(np.asarray(B_c)); d = _lib.f64(np.asarray(d_c))
        Bn, n, m = A.shape[0], A.shape[-1], Bm.shape[-1]
        Ad = np.empty((Bn, n, n)); Bd = np.empty((Bn, n, m)); dd = np.empty((Bn, n))
        _lib.check(_lib.lib().stpwl_discretize(C.c_int(methods[self.discr_method]), C.c_int(n), C.c_int(m), C.c_int64(Bn), _lib.dptr(A),
                                               _lib.dptr(Bm), _lib.dptr(d), C.c_double(float(dt)), _lib.dptr(Ad), _lib.dptr(Bd),
                                               _lib.dptr(dd)), 'stpwl_discretize')
        return Ad, Bd, dd

    def discretize_dynamics(self, A_c, B_c, d_c, dt):
        """tpwl.py:272-297 for one model (on the device: discretize_batch)."""
        A_d, B_d, d_d = self.discretize_batch(np.asarray(A_c)[None], np.asarray(B_c)[None], np.asarray(d_c)[None], dt)
        return A_d[0], B_d[0], d_d[0]

    def pre_discretize(self, dt):
        """tpwl.py:299-322: discretise all stored points, keep them as lists (reference attribute
        layout) and install the tables on the device."""
        if self.tpwl_method != 'nn':
            raise RuntimeError('tpwl method should be nn to pre-discretize')
        print('Performing pre-discretization using {} of TPWL model with dt = {:.3f}'
              .format(DISCR_DICT[self.discr_method], dt))
        Ac, Bc, dc = self._tabs[3], self._tabs[4], self._tabs[5]
        A_d, B_d, d_d = self.discretize_batch(Ac, Bc, dc, dt)           # all stored points in one launch
        self.A_d, self.B_d, self.d_d = list(A_d), list(B_d), list(d_d)
        if float(dt) not in self._dt_handles:          # (a handle created earlier for this dt holds the same tables)
            self.handle_for(dt, tables=(self.A_d, self.B_d, self.d_d))
        self.pre_discretized_dt = dt

    def _ensure_discrete(self, dt):
        """Make sure a device handle with the tables discretised at dt exists (see handle_for)."""
        self.handle_for(dt)

    def get_characteristic_dx(self, dt):
        """tpwl.py:324-334."""
        x = scutils.qv2x(self._tabs[0], self._tabs[1])
        if self.tpwl_method == 'nn':
            self._ensure_discrete(dt)
        A, B, d, _ = self.linearize_batch(x, dt)
        dx = np.einsum('bij,bj->bi', A, x) + np.einsum('bij,bj->bi', B, self._tabs[2]) + d - x
        return np.abs(dx).max(axis=0)

    @staticmethod
    def update_dynamics(x, u, A_d, B_d, d_d):
        """tpwl.py:336-339 (host helper on caller-supplied arrays)."""
        return A_d @ x + np.squeeze(B_d @ u) + d_d

    def get_ref_point(self):
        return self.ref_point
