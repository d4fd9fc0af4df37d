"""Host-side helpers with the reference's names (sofacontrol/utils.py): state packing, cost and
polyhedron containers, list<->array wire helpers.  Pure data handling, no arithmetic hot path."""
import os
import pickle

import numpy as np


class QuadraticCost:
    """sofacontrol/utils.py:8-16."""

    def __init__(self, Q=None, R=None, Qf=None):
        self.Qf = Qf
        self.Q = Q
        self.R = R


def qv2x(q, v):
    """x = [v; q] (sofacontrol/utils.py:135-136)."""
    return np.concatenate((v, q), axis=-1)


def x2qv(x):
    """Returns (q, v) (sofacontrol/utils.py:139-148)."""
    if x.ndim == 1:
        n = x.shape[0] // 2
        return x[n:], x[:n]
    if x.ndim == 2:
        n = x.shape[-1] // 2
        return x[:, n:], x[:, :n]
    raise IndexError('Unable to process x.ndim > 2')


def vq2qv(x):
    q, v = x2qv(x)
    return np.hstack((q, v))


def save_data(filename, data):
    d = os.path.split(filename)[0]
    if d and not os.path.isdir(d):
        os.mkdir(d)
    with open(filename, 'wb') as f:
        pickle.dump(data, f, protocol=pickle.HIGHEST_PROTOCOL)


def load_data(filename):
    with open(filename, 'rb') as f:
        return pickle.load(f)


class Polyhedron:
    """{x : A x <= b} (sofacontrol/utils.py:364-407).  with_reproject=True enables `project_to_polyhedron`: the
    Euclidean projection min 1/2 |p - x|^2 s.t. A p <= b -- the QP the reference hands to OSQP -- solved exactly on
    the device (`spoly_project`, csrc/poly.hip)."""

    def __init__(self, A, b, with_reproject=False):
        self.A = np.asarray(A, dtype=np.float64)
        self.b = np.asarray(b, dtype=np.float64)
        self.with_reproject = with_reproject

    def contains(self, x):
        return not (np.max(self.A @ x - self.b) > 0)

    def get_constraint_violation(self, x):
        return np.linalg.norm(np.maximum(self.A @ x - self.b, 0))

    def project_to_polyhedron(self, x):
        if not self.with_reproject:
            raise RuntimeError('Reproject not specified for class instance, set with_reproject=True to enable'
                               'reprojection to the Polyhedron through a QP')
        import ctypes as C
        from . import _lib
        X = _lib.f64(np.atleast_2d(x))
        A, b = _lib.f64(self.A), _lib.f64(self.b)
        if X.shape[1] != A.shape[1]:
            raise RuntimeError('project_to_polyhedron: expected points of dimension %d, got %s' % (A.shape[1], np.shape(x)))
        out = np.empty_like(X)
        _lib.check(_lib.lib().spoly_project(_lib.dptr(A), _lib.dptr(b), C.c_int(A.shape[0]), C.c_int(A.shape[1]),
                                            _lib.dptr(X), C.c_int64(X.shape[0]), _lib.dptr(out)), 'spoly_project')
        return out[0] if np.ndim(x) == 1 else out


class HyperRectangle(Polyhedron):
    """Rows alternate +e_i, -e_i with b = [ub_i, -lb_i, ...] (sofacontrol/utils.py:409-414)."""

    def __init__(self, ub, lb):
        n = len(ub)
        A = np.kron(np.eye(n), np.array([[1.], [-1.]]))
        b = np.hstack([np.array([ub[i], -lb[i]]) for i in range(n)])
        super().__init__(A, b)


def arr2np(x, dim, squeeze=False):
    """sofacontrol/utils.py:417-425."""
    a = np.asarray(x, dtype='float64').reshape(-1, dim)
    return a.squeeze() if squeeze else a


def np2arr(x):
    """sofacontrol/utils.py:428-431."""
    return x.flatten().tolist()


class Point:
    """sofacontrol/utils.py:19-39."""

    def __init__(self):
        self.step = None
        self.t = None
        self.q = None
        self.v = None
        self.u = None
        self.H = None
        self.K = None
        self.D = None
        self.M = None
        self.S = None
        self.f = None
        self.b = None
        self.q_next = None
        self.v_next = None
        self.dt = None


class SnapshotData:
    """sofacontrol/utils.py:42-107: container filled by the open-loop data collection."""

    def __init__(self, save_dynamics=True):
        self.save_dynamics = save_dynamics
        keys = ['t', 'q', 'v', 'u'] + (['H', 'K', 'D', 'M', 'S', 'b', 'f'] if save_dynamics else []) + ['q+', 'v+']
        self.dict = {k: [] for k in keys}
        self.dict['dt'] = -1

    def add_point(self, point):
        if self.dict['dt'] == -1:
            self.dict['dt'] = point.dt
        self.dict['t'].append(point.t)
        self.dict['q'].append(point.q)
        self.dict['v'].append(point.v)
        self.dict['u'].append(point.u)
        self.dict['q+'].append(point.q_next)
        self.dict['v+'].append(point.v_next)
        if self.save_dynamics:
            for k in ('K', 'D', 'M', 'b', 'f', 'H', 'S'):
                self.dict[k].append(getattr(point, k))

    def save_snapshot(self, *args):
        return True

    def simulation_end(self, filename):
        dict_lists_to_array(self.dict)
        save_data(filename, self.dict)


def dict_lists_to_array(d):
    """sofacontrol/utils.py:338-344."""
    for key in d:
        if type(d[key]) == list:
            d[key] = np.asarray(d[key])


def extract_AB(K, D, M, H):
    """sofacontrol/utils.py:251-284 for reduced (dense r x r) matrices: A = [[-M^-1 D, -M^-1 K], [I, 0]],
    B = [[M^-1 H], [0]].  One-off r x r algebra per TPWL point; the O(n_f^2) work is the U^T . U reduction that
    produced K, D, M (POD.compute_RO_matrix on the device)."""
    K, D, M, H = (np.asarray(a.toarray() if hasattr(a, 'toarray') else a, dtype=np.float64) for a in (K, D, M, H))
    Minv = np.linalg.inv(M)
    A11, A12, Ht = -(Minv @ D), -(Minv @ K), Minv @ H
    A = np.block([[A11, A12], [np.eye(A11.shape[0]), np.zeros(A12.shape)]])
    B = np.block([[Ht], [np.zeros(Ht.shape)]])
    return A, B


def extract_AB_d(S, K, H, dt):
    """sofacontrol/utils.py:287-299 (ThieffryKruszewskiEtAl2019 discrete derivation)."""
    Sinv = np.linalg.inv(S)
    SinvK, SinvH = Sinv @ K, Sinv @ H
    dim = K.shape[0]
    I = np.eye(dim)
    A = np.block([[I - dt ** 2 * SinvK, -dt * SinvK], [dt * I - dt ** 3 * SinvK, I - dt ** 2 * SinvK]])
    B = np.block([[dt * SinvH], [dt ** 2 * SinvH]])
    return A, B


def zoh_linear(A, B, dt):
    """sofacontrol/utils.py:302-320: exact zero-order-hold discretisation, expm([[A, B], [0, 0]] dt) -- on the device
    (csrc/discretize.hip: scaling and squaring, [13/13] Pade approximant)."""
    A_d, B_d, _ = zoh_affine(A, B, np.zeros(np.shape(A)[0]), dt)
    return A_d, B_d


def zoh_affine(A, B, d, dt):
    """sofacontrol/utils.py:323-335: expm([[A, B, d], [0, 0, 0]] dt); the affine term rides along as one more input column."""
    import ctypes as C
    from . import _lib
    A = _lib.f64(np.asarray(A)); B = _lib.f64(np.asarray(B)); d = _lib.f64(np.asarray(d).reshape(-1))
    n, m = B.shape
    Ad = np.empty((n, n)); Bd = np.empty((n, m)); dd = np.empty(n)
    _lib.check(_lib.lib().stpwl_discretize(C.c_int(3), C.c_int(n), C.c_int(m), C.c_int64(1), _lib.dptr(A), _lib.dptr(B), _lib.dptr(d),
                                           C.c_double(float(dt)), _lib.dptr(Ad), _lib.dptr(Bd), _lib.dptr(dd)), 'stpwl_discretize')
    return Ad, Bd, dd


def sparse_list_to_np_array(matrix_list):
    """sofacontrol/utils.py:162-163."""
    return np.asarray([np.asarray(matrix.todense()) for matrix in matrix_list])
