"""SSM polynomial reduced model (sofacontrol/SSM/ssm.py) over the HIP library (csrc/ssm.hip).

Same constructor (`.mat`-style nested `model` / `params` dictionaries), attribute and method names as the
reference; the polynomial maps, their Jacobians (analytic, the reference uses jax), the discretisation
(fe / be / bil) and the rollout run on the device in float64."""
import ctypes as C

import numpy as np

from .. import _lib

DISCR_METHOD = 'zoh'
TPWL_METHOD = 'nn'
DISCR_DICT = {'fe': 'forward Euler', 'be': 'implicit Euler', 'bil': 'bilinear transform', 'zoh': 'zero-order hold'}
_MODES = {'fe': 1, 'be': 2, 'bil': 3}
_CONT, _DISCRETE_MAP = 0, 4


def _i32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class SSM:
    """ssm.py:18-178."""

    def __init__(self, eq_point, discrete=False, discr_method='fe', **kwargs):
        self.maps = {}
        self.discrete = discrete
        self.discr_method = discr_method
        self.model = kwargs.pop('model', None)
        self.params = kwargs.pop('params', None)
        self.state_dim = int(self.params['state_dim'][0, 0][0, 0])
        self.input_dim = int(self.params['input_dim'][0, 0][0, 0])
        self.output_dim = int(self.params['output_dim'][0, 0][0, 0])
        self.SSM_order = int(self.params['SSM_order'][0, 0][0, 0])
        self.ROM_order = int(self.params['ROM_order'][0, 0][0, 0])
        self.Ts = self.model['Ts'][0, 0][0, 0]
        self.w_coeff = _lib.f64(self.model['w_coeff'][0, 0])
        self.v_coeff = _lib.f64(self.model['v_coeff'][0, 0])
        self.r_coeff = _lib.f64(self.model['r_coeff'][0, 0])
        self.B_r = _lib.f64(self.model['B'][0, 0])
        self.rd_coeff = _lib.f64(self.model['rd_coeff'][0, 0])
        self.Bd_r = _lib.f64(self.model['Bd'][0, 0])
        self.z_ref = _lib.f64(eq_point)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().sssm_create(C.byref(self._h), C.c_int(self.state_dim), C.c_int(self.input_dim),
                                          C.c_int(self.output_dim), C.c_int(self.ROM_order), C.c_int(self.SSM_order),
                                          _lib.dptr(self.r_coeff), _lib.dptr(self.B_r), _lib.dptr(self.rd_coeff),
                                          _lib.dptr(self.Bd_r), _lib.dptr(self.w_coeff), _lib.dptr(self.v_coeff),
                                          _lib.dptr(self.z_ref)), 'sssm_create')
        self.rom_phi = self.get_poly_basis(self.state_dim, self.ROM_order)
        self.ssm_phi = self.get_poly_basis(self.output_dim, self.SSM_order)
        self.C_map = self.reduced_to_observed
        self.W_map = self.observed_to_reduced
        self.maps['f_nl'] = self.reduced_dynamics
        if self.discrete:
            self.maps['f_nl_d'] = self.reduced_dynamics_discrete
        self.A_d = None
        self.B_d = None
        self.d_d = None
        self.H = np.zeros((self.output_dim, self.state_dim))
        self.nonlinear_observer = True

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            try:
                _lib.lib().sssm_destroy(h)
            except Exception:
                pass
            self._h = None

    @property
    def handle(self):
        return self._h

    def update_state(self, x, u, dt):
        raise NotImplementedError("update_state must be overriden by a child class")

    def get_jacobians(self, x, u, dt):
        raise NotImplementedError("get_jacobians must be overriden by a child class")

    def zfyf_to_zy(self, zf=None):
        if zf is not None and self.z_ref is not None:
            return zf - self.z_ref
        raise RuntimeError('Need to specify equilibrium point')

    def zy_to_zfyf(self, z=None):
        if z is not None and self.z_ref is not None:
            return z + self.z_ref
        raise RuntimeError('Need to specify equilibrium point')

    def x_to_zfyf(self, x, zf=True):
        """ssm.py:101-107: C_map(x) + z_ref for x (N, n_x) or (n_x,)."""
        return self.C_map(np.asarray(x).T).T + self.z_ref

    def x_to_zy(self, x):
        return self.C_map(x)

    def get_sim_params(self):
        """ssm.py:121-123 (the reference's dict names two attributes an SSM model does not have; the discretisation is
        what identifies the run)."""
        return {'discr_method': self.discr_method, 'discrete': self.discrete}

    def get_state_dim(self):
        return self.state_dim

    def get_input_dim(self):
        return self.input_dim

    def get_output_dim(self):
        return self.output_dim

    def _mode(self):
        if self.discrete:
            return _DISCRETE_MAP
        if self.discr_method not in _MODES:
            raise RuntimeError('self.discr_method must be in [fe, be, bil, zoh]')
        return _MODES[self.discr_method]

    def rollout(self, x0, u, dt):
        """ssm.py:134-156: x0 (n_x,), u (N, n_u) -> x (N+1, n_x), z (N+1, n_z).  Batched: x0 (B, n_x),
        u (B, N, n_u)."""
        x0a = np.ascontiguousarray(np.atleast_2d(x0), dtype=np.float64)
        ua = np.ascontiguousarray(u, dtype=np.float64)
        single = ua.ndim == 2
        if single:
            ua = ua[None]
        Bn, N = ua.shape[0], ua.shape[1]
        X = np.empty((Bn, N + 1, self.state_dim))
        Z = np.empty((Bn, N + 1, self.output_dim))
        _lib.check(_lib.lib().sssm_rollout(self._h, _lib.dptr(x0a), _lib.dptr(ua), C.c_int(N), C.c_int64(Bn),
                                           C.c_int(self._mode()), C.c_double(dt), _lib.dptr(X), _lib.dptr(Z)),
                   'sssm_rollout')
        return (X[0], Z[0]) if single else (X, Z)

    def get_poly_basis(self, dim, order):
        """ssm.py:158-164: callable phi(*x) -> list of monomials (graded, lexicographic with x1 first)."""
        nm = _lib.lib().sssm_num_monomials(C.c_int(dim), C.c_int(order))
        E = np.empty((nm, dim), dtype=np.int32)
        _lib.check(_lib.lib().sssm_exponents(C.c_int(dim), C.c_int(order), _i32(E)), 'sssm_exponents')

        def phi(*x):
            xv = np.asarray(x)
            return [np.prod([xv[i] ** int(e[i]) for i in range(dim) if e[i]], axis=0) for e in E]
        phi.exponents = E
        return phi

    # ---- maps (device)
    def _dyn(self, x, u, discrete):
        X = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        U = np.ascontiguousarray(np.atleast_2d(u), dtype=np.float64)
        F = np.empty_like(X)
        _lib.check(_lib.lib().sssm_dynamics(self._h, _lib.dptr(X), _lib.dptr(U), C.c_int64(X.shape[0]),
                                            C.c_int(discrete), _lib.dptr(F)), 'sssm_dynamics')
        return F[0] if np.ndim(x) == 1 else F

    def reduced_dynamics(self, x, u):
        return self._dyn(x, u, 0)

    def reduced_dynamics_discrete(self, x, u):
        return self._dyn(x, u, 1)

    def reduced_to_observed(self, x):
        """C_map (ssm.py:170-171): x (n_x,) or (n_x, N) column-stacked -> z without z_ref, same layout."""
        xa = np.asarray(x, dtype=np.float64)
        X = np.ascontiguousarray(xa.T if xa.ndim == 2 else xa[None], dtype=np.float64)
        Z = np.empty((X.shape[0], self.output_dim))
        _lib.check(_lib.lib().sssm_observe(self._h, _lib.dptr(X), C.c_int64(X.shape[0]), _lib.dptr(Z), None, None),
                   'sssm_observe')
        return Z.T if xa.ndim == 2 else Z[0]

    def observed_to_reduced(self, z):
        """W_map (ssm.py:173-174) of the already shifted observation z (n_z,) or (n_z, N)."""
        za = np.asarray(z, dtype=np.float64)
        Zs = np.ascontiguousarray((za.T if za.ndim == 2 else za[None]) + self.z_ref, dtype=np.float64)
        X = np.empty((Zs.shape[0], self.state_dim))
        _lib.check(_lib.lib().sssm_reduce(self._h, _lib.dptr(Zs), C.c_int64(Zs.shape[0]), _lib.dptr(X)),
                   'sssm_reduce')
        return X.T if za.ndim == 2 else X[0]


class SSMDynamics(SSM):
    """ssm.py:181-344."""

    def update_state(self, x, u, dt):
        A_d, B_d, d_d = self.get_jacobians(x, dt=dt, u=u)
        return self.update_dynamics(x, u, A_d, B_d, d_d)

    def _lin(self, x, u, mode, dt):
        X = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        U = np.ascontiguousarray(np.atleast_2d(u), dtype=np.float64)
        Bn, n, m = X.shape[0], self.state_dim, self.input_dim
        A = np.empty((Bn, n, n)); B = np.empty((Bn, n, m)); d = np.empty((Bn, n))
        _lib.check(_lib.lib().sssm_linearize(self._h, _lib.dptr(X), _lib.dptr(U), C.c_int64(Bn), C.c_int(mode),
                                             C.c_double(0.0 if dt is None else dt), _lib.dptr(A), _lib.dptr(B),
                                             _lib.dptr(d)), 'sssm_linearize')
        return (A[0], B[0], d[0]) if np.ndim(x) == 1 else (A, B, d)

    def get_continuous_jacobians(self, x, u):
        """ssm.py:198-204."""
        return self._lin(x, u, _CONT, None)

    def get_discrete_jacobians(self, x, u):
        """ssm.py:206-212."""
        return self._lin(x, u, _DISCRETE_MAP, None)

    def get_jacobians(self, x, u, dt):
        """ssm.py:214-218; x, u may be batches (B, n_x), (B, n_u)."""
        return self._lin(x, u, self._mode(), dt)

    def get_observer_jacobians(self, x):
        """ssm.py:220-227: H = dC/dx, c = C(x) - H x."""
        X = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        Bn = X.shape[0]
        H = np.empty((Bn, self.output_dim, self.state_dim)); c = np.empty((Bn, self.output_dim))
        _lib.check(_lib.lib().sssm_observe(self._h, _lib.dptr(X), C.c_int64(Bn), None, _lib.dptr(H), _lib.dptr(c)),
                   'sssm_observe')
        return (H[0], c[0]) if np.ndim(x) == 1 else (H, c)

    def update_observer_state(self, x, dt=None, u=None):
        H, c = self.get_observer_jacobians(x)
        return np.squeeze(H @ x) + np.squeeze(c)

    def discretize_dynamics(self, A_c, B_c, d_c, dt):
        """ssm.py:279-301 on caller-supplied matrices (csrc/discretize.hip; the model's own path discretises in-kernel)."""
        methods = {'fe': 0, 'be': 1, 'bil': 2}
        if self.discr_method not in methods:
            raise RuntimeError('self.discr_method must be in [fe, be, bil, zoh]')
        A = _lib.f64(np.asarray(A_c)); B = _lib.f64(np.asarray(B_c)); d = _lib.f64(np.asarray(d_c).reshape(-1))
        n, m = B.shape
        Ad = np.empty((n, n)); Bd = np.empty((n, m)); dd = np.empty(n)
        _lib.check(_lib.lib().stpwl_discretize(C.c_int(methods[self.discr_method]), C.c_int(n), C.c_int(m), C.c_int64(1), _lib.dptr(A),
                                               _lib.dptr(B), _lib.dptr(d), C.c_double(float(dt)), _lib.dptr(Ad), _lib.dptr(Bd),
                                               _lib.dptr(dd)), 'stpwl_discretize')
        return Ad, Bd, dd

    @staticmethod
    def update_dynamics(x, u, A_d, B_d, d_d):
        return np.squeeze(A_d @ x) + np.squeeze(B_d @ u) + np.squeeze(d_d)

    def get_ref_point(self):
        return self.z_ref

    def compute_RO_state(self, z):
        """ssm.py:338-344: W_map(z - z_ref); z (n_z,) or (B, n_z) rows."""
        Z = np.ascontiguousarray(np.atleast_2d(z), dtype=np.float64)
        X = np.empty((Z.shape[0], self.state_dim))
        _lib.check(_lib.lib().sssm_reduce(self._h, _lib.dptr(Z), C.c_int64(Z.shape[0]), _lib.dptr(X)), 'sssm_reduce')
        return X[0] if np.ndim(z) == 1 else X
