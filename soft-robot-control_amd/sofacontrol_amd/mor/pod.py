"""POD reduced-order map on MI355X: same surface as sofacontrol/mor/pod.py (class POD, pod_config,
load_POD, run_POD, get_snapshots, process_snapshots, compute_POD), arithmetic in HIP kernels."""
import ctypes as C
import os

import numpy as np

from .. import _lib
from .. import utils as scutils

SROM_Q, SROM_V, SROM_X, SROM_RAW = 0, 1, 2, 3


class POD:
    """POD object (sofacontrol/mor/pod.py:9-78).

    `compute_RO_state` / `compute_FO_state` accept one vector (the reference's use) or a
    (B, n) batch with one snapshot per row (the layout of np.asarray(data['q']), pod.py:149).
    """

    def __init__(self, POD_info):
        self.q_ref = np.asarray(POD_info['q_ref'], dtype=np.float64)
        self.v_ref = np.asarray(POD_info['v_ref'], dtype=np.float64)
        self.x_ref = scutils.qv2x(self.q_ref, self.v_ref)
        self.U = np.ascontiguousarray(POD_info['U'], dtype=np.float64)
        self.rom_dim = self.U.shape[1]
        self._V = None
        self._h = C.c_void_p()
        lib = _lib.lib()
        _lib.check(lib.srom_create(C.byref(self._h), _lib.dptr(self.U), C.c_int64(self.U.shape[0]),
                                   C.c_int(self.rom_dim), _lib.dptr(_lib.f64(self.q_ref)),
                                   _lib.dptr(_lib.f64(self.v_ref))), 'srom_create')

    def __del__(self):
        try:
            if self._h:
                _lib.lib().srom_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    @property
    def V(self):
        """V = kron(I2, U) (pod.py:19); materialised lazily -- the kernels use the block structure."""
        if self._V is None:
            self._V = np.kron(np.eye(2), self.U)
        return self._V

    @property
    def handle(self):
        return self._h

    def _apply(self, fn, which, a, n_in, n_out, name):
        a = np.asarray(a, dtype=np.float64)
        single = a.ndim == 1
        A = np.ascontiguousarray(a.reshape(1, -1) if single else a)
        if A.ndim != 2 or A.shape[1] != n_in:
            raise RuntimeError('%s: expected trailing dimension %d, got %s' % (name, n_in, a.shape))
        out = np.empty((A.shape[0], n_out))
        _lib.check(fn(self._h, C.c_int(which), _lib.dptr(A), C.c_int64(A.shape[0]), _lib.dptr(out)), name)
        return out[0] if single else out

    def compute_FO_state(self, q=None, v=None, x=None):
        """pod.py:22-37."""
        n_f, r = self.U.shape
        lift = _lib.lib().srom_lift
        if q is not None:
            return self._apply(lift, SROM_Q, q, r, n_f, 'srom_lift')
        elif v is not None:
            return self._apply(lift, SROM_V, v, r, n_f, 'srom_lift')
        elif x is not None:
            return self._apply(lift, SROM_X, x, 2 * r, 2 * n_f, 'srom_lift')
        raise RuntimeError('Must specify vector type')

    def compute_RO_state(self, qf=None, vf=None, xf=None):
        """pod.py:39-54."""
        n_f, r = self.U.shape
        proj = _lib.lib().srom_project
        if qf is not None:
            return self._apply(proj, SROM_Q, qf, n_f, r, 'srom_project')
        elif vf is not None:
            return self._apply(proj, SROM_V, vf, n_f, r, 'srom_project')
        elif xf is not None:
            return self._apply(proj, SROM_X, xf, 2 * n_f, 2 * r, 'srom_project')
        raise RuntimeError('Must specify vector type')

    def compute_RO_matrix(self, matrix, left=False, right=False):
        """pod.py:56-72 (ndarray or scipy coo_matrix; coo is densified as the reference's K, D, M, S
        are created from dense arrays, utils.py:187-206)."""
        try:
            from scipy.sparse import coo_matrix
            sparse_ok = isinstance(matrix, coo_matrix)
        except Exception:
            sparse_ok = False
        if not (isinstance(matrix, np.ndarray) or sparse_ok):
            raise RuntimeError('Matrix is not numpy ndarray or sparse coo_matrix')
        M = matrix.toarray() if sparse_ok else matrix
        vec = (M.ndim == 1)
        M = np.ascontiguousarray(M.reshape(-1, 1) if vec else M, dtype=np.float64)
        n_f, r = self.U.shape
        if M.shape[0] != n_f:
            raise RuntimeError('matrix must have n_f rows')
        both = (left and right) or (not left and not right)
        if both:
            out = np.empty((r, r))
        elif left:
            out = np.empty((r, M.shape[1]))
        else:
            out = np.empty((n_f, r))
        _lib.check(_lib.lib().srom_reduce_matrix(self._h, _lib.dptr(M), C.c_int64(M.shape[1]),
                                                 C.c_int(bool(left)), C.c_int(bool(right)),
                                                 _lib.dptr(out)), 'srom_reduce_matrix')
        return out[:, 0] if (vec and left and not right) else out

    def compute_RO_matrices(self, matrices):
        """U^T M U (pod.py:56-72 with left = right = True) for several n_f x n_f matrices in one call: TPWLSnapshotData.add_point
        reduces K, D, M and S of one linearisation point back to back (tpwl/tpwl_utils.py:96-103) -- groups of four share one
        launch pair (srom_reduce_matrices).  Returns a list of r x r arrays, one per input, same arithmetic as compute_RO_matrix."""
        try:
            from scipy.sparse import coo_matrix
        except Exception:
            coo_matrix = ()
        n_f, r = self.U.shape
        dense = []
        for matrix in matrices:
            if not (isinstance(matrix, np.ndarray) or (coo_matrix and isinstance(matrix, coo_matrix))):
                raise RuntimeError('Matrix is not numpy ndarray or sparse coo_matrix')
            M = np.ascontiguousarray(matrix.toarray() if not isinstance(matrix, np.ndarray) else matrix, dtype=np.float64)
            if M.shape != (n_f, n_f):
                raise RuntimeError('matrix must be n_f x n_f')
            dense.append(M)
        outs = [np.empty((r, r)) for _ in dense]
        if dense:
            PP = C.POINTER(C.c_double) * len(dense)
            _lib.check(_lib.lib().srom_reduce_matrices(self._h, PP(*[_lib.dptr(M) for M in dense]), C.c_int(len(dense)),
                                                       PP(*[_lib.dptr(o) for o in outs])), 'srom_reduce_matrices')
        return outs

    def get_info(self):
        """pod.py:74-78."""
        return {'q_ref': self.q_ref, 'v_ref': self.v_ref, 'U': self.U, 'type': 'POD'}


class pod_config():
    """pod.py:81-90."""

    def __init__(self):
        self.pod_type = 'v'
        self.pod_tolerance = 0.0001
        self.preprocess = []
        self.preprocess_args = {'nbr_clusters': 0}


def load_POD(POD_file):
    """pod.py:93-107."""
    if not os.path.isfile(POD_file):
        raise RuntimeError('POD file specified is not a valid file')
    POD_data = scutils.load_data(POD_file)
    return POD(POD_data['POD_info'])


def get_snapshots(data, pod_type):
    """pod.py:144-154."""
    if pod_type == 'q':
        return np.asarray(data['q']) - data['q'][0]
    elif pod_type == 'v':
        return np.asarray(data['v'])
    elif pod_type == 'a':
        return np.asarray(data['v+']) - np.asarray(data['v'])
    raise RuntimeError('pod_type must be q, v or a')


def process_snapshots(snapshots, preprocess, args):
    """pod.py:157-178 on the device: the snapshot matrix (n_s x n_f, one snapshot per row) is uploaded once, 'normalize',
    'substract_mean' and 'clustering' run on the resident copy (csrc/snapshots.hip), the result comes back as an array."""
    wanted = [p_ for p_ in ('normalize', 'substract_mean', 'clustering') if p_ in preprocess]
    if 'clustering' in wanted and not args.get('nbr_clusters', 0) > 0:
        print('Not using kmeans because nbr_clusters not specified in config.preprocess_args dictionary')
        wanted.remove('clustering')
    if not wanted:
        return snapshots
    S = np.ascontiguousarray(snapshots, dtype=np.float64)
    dS, n_s, n_f = _process_snapshots_dev(_lib.DeviceBuffer.from_array(S), S.shape[0], S.shape[1], wanted, args)
    return dS.to_array((n_s, n_f))


def _process_snapshots_dev(dS, n_s, n_f, wanted, args):
    """The preprocessing steps on a resident snapshot matrix; returns (buffer, n_s, n_f) -- clustering replaces the
    snapshots by the k centroids."""
    L = _lib.lib()
    n64 = (C.c_int64(n_s), C.c_int64(n_f), C.c_int64(n_f))
    if 'normalize' in wanted:
        dmin, dmax = _lib.DeviceBuffer(n_f * 8), _lib.DeviceBuffer(n_f * 8)
        _lib.check(L.srom_snapshot_stats_dev(dS.ptr, *n64, dmin.ptr, dmax.ptr, None, None), 'srom_snapshot_stats_dev')
        _lib.check(L.srom_snapshot_normalize_dev(dS.ptr, *n64, dmin.ptr, dmax.ptr, None), 'srom_snapshot_normalize_dev')
    if 'substract_mean' in wanted:
        dmean = _lib.DeviceBuffer(n_f * 8)
        _lib.check(L.srom_snapshot_stats_dev(dS.ptr, *n64, None, None, dmean.ptr, None), 'srom_snapshot_stats_dev')
        _lib.check(L.srom_snapshot_center_dev(dS.ptr, *n64, dmean.ptr, None), 'srom_snapshot_center_dev')
    if 'clustering' in wanted:
        k = int(args['nbr_clusters'])
        print('Computing %d centroids for an initial snapshot size of %d using k-means clustering' % (k, n_s))
        dS = _kmeans_centroids_dev(dS, n_s, n_f, k)
        n_s = k
    _lib.sync()
    return dS, n_s, n_f


def compute_kmeans_centroids(snapshot, k, n_init=100, max_iter=1000, random_state=0, tol=1e-4):
    """pod.py:207-216: the centroids sklearn's KMeans(k, n_init=100, max_iter=1000, random_state=0) returns, with the
    distance products, assignments and centroid sums on the device (see _kmeans_centroids_dev)."""
    S = np.ascontiguousarray(snapshot, dtype=np.float64)
    print('Computing %d centroids for an initial snapshot size of %d using k-means clustering' % (k, S.shape[0]))
    dC = _kmeans_centroids_dev(_lib.DeviceBuffer.from_array(S), S.shape[0], S.shape[1], int(k), n_init, max_iter, random_state, tol)
    return dC.to_array((int(k), S.shape[1]))


def _same_clustering(a, b, k):
    """sklearn's _is_same_clustering: equal up to a permutation of the labels."""
    mapping = np.full(k, -1, dtype=np.int64)
    for la, lb in zip(a, b):
        if mapping[la] == -1:
            mapping[la] = lb
        elif mapping[la] != lb:
            return False
    return True


def _kmeans_centroids_dev(dS, n_s, n_f, k, n_init=100, max_iter=1000, random_state=0, tol=1e-4):
    """scikit-learn's KMeans.fit (dense Lloyd, k-means++ seeding) restated around device kernels.  The data are centred
    first (as the estimator does, for the accuracy of the distance products), every one of the n_init runs is seeded by
    k-means++ with draws from ONE numpy RandomState(random_state) in the estimator's order (`choice` for the first
    centre, 2 + int(log k) `uniform` trials per further centre), and the best run is the first with the smallest inertia
    whose clustering differs.  The random draws, the cumulative sums they are compared with and the bookkeeping are host
    work on n_s-vectors; every pass over the n_s x n_f data runs on the device."""
    L = _lib.lib()
    n64 = (C.c_int64(n_s), C.c_int64(n_f), C.c_int64(n_f))
    dX = _lib.DeviceBuffer(n_s * n_f * 8)
    _lib.check(L.srh_memcpy_d2d(dX.ptr, dS.ptr, C.c_size_t(n_s * n_f * 8)), 'srh_memcpy_d2d')
    dmean, dxn = _lib.DeviceBuffer(n_f * 8), _lib.DeviceBuffer(n_s * 8)
    _lib.check(L.srom_snapshot_stats_dev(dX.ptr, *n64, None, None, dmean.ptr, None), 'srom_snapshot_stats_dev')
    _lib.check(L.srom_snapshot_center_dev(dX.ptr, *n64, dmean.ptr, None), 'srom_snapshot_center_dev')
    _lib.check(L.srom_row_sqnorms_dev(dX.ptr, *n64, dxn.ptr, None), 'srom_row_sqnorms_dev')
    _lib.sync()
    xn = dxn.to_array((n_s,))
    tol_abs = float(xn.sum() / (n_s * n_f)) * tol          # mean column variance of the centred data x tol
    rs = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
    n_trials = 2 + int(np.log(k))
    dY, dD = _lib.DeviceBuffer(n_trials * n_f * 8), _lib.DeviceBuffer(n_trials * n_s * 8)
    dC, dlab = _lib.DeviceBuffer(k * n_f * 8), _lib.DeviceBuffer(n_s * 4)
    row_bytes = n_f * 8

    def rows_to(dst, ids):
        for t, i in enumerate(ids):
            _lib.check(L.srh_memcpy_d2d(C.c_void_p(dst.ptr.value + t * row_bytes), C.c_void_p(dX.ptr.value + int(i) * row_bytes),
                                        C.c_size_t(row_bytes)), 'srh_memcpy_d2d')

    def dist_rows(ids):
        rows_to(dY, ids)
        _lib.check(L.srom_sqdist_rows_dev(dX.ptr, *n64, dY.ptr, C.c_int(len(ids)), dxn.ptr, dD.ptr, None), 'srom_sqdist_rows_dev')
        return dD.to_array((n_trials, n_s))[:len(ids)]

    best = None
    for _ in range(n_init):
        # k-means++ (sklearn _kmeans_plusplus, unit sample weights)
        indices = np.empty(k, dtype=np.int64)
        indices[0] = rs.choice(n_s, p=np.full(n_s, 1.0 / n_s))
        closest = dist_rows([indices[0]])[0].copy()
        pot = closest.sum()
        for c in range(1, k):
            rand_vals = rs.uniform(size=n_trials) * pot
            cand = np.searchsorted(np.cumsum(closest, dtype=np.float64), rand_vals)
            np.clip(cand, None, n_s - 1, out=cand)
            d = np.minimum(closest, dist_rows(cand))
            pots = d.sum(axis=1)
            b = int(np.argmin(pots))
            pot, closest, indices[c] = pots[b], d[b].copy(), cand[b]
        rows_to(dC, indices)
        inertia, iters = C.c_double(), C.c_int()
        _lib.check(L.srom_kmeans_lloyd_dev(dX.ptr, *n64, C.c_int(k), dC.ptr, C.c_int(max_iter), C.c_double(tol_abs), dlab.ptr,
                                           C.byref(inertia), C.byref(iters), None), 'srom_kmeans_lloyd_dev')
        labels = dlab.to_array((n_s,), dtype=np.int32)
        if best is None or (inertia.value < best[0] and not _same_clustering(labels, best[1], k)):
            best = (inertia.value, labels, dC.to_array((k, n_f)))
    centers = best[2] + dmean.to_array((n_f,))
    return _lib.DeviceBuffer.from_array(np.ascontiguousarray(centers))


def energy_truncation(S, tol):
    """pod.py:192-197: smallest k >= 1 with sum(S[k:]^2) / sum(S^2) <= tol."""
    s_square = S ** 2
    i = 0
    while (np.sum(s_square[i:]) / np.sum(s_square)) > tol or i == 0:
        i += 1
    return i


def gramian(S_rows):
    """G = S S^T on the device for S (n_s x n_f) with one snapshot per row."""
    S_rows = np.ascontiguousarray(S_rows, dtype=np.float64)
    n_s, n_f = S_rows.shape
    G = np.empty((n_s, n_s))
    _lib.check(_lib.lib().srom_gramian(_lib.dptr(S_rows), C.c_int64(n_s), C.c_int64(n_f), _lib.dptr(G)), 'srom_gramian')
    return G


def _device_eigh(G):
    """(w ascending, W with eigenvectors as columns) of the symmetric G through srom_eigh_dev."""
    n_s = G.shape[0]
    dG, dw = _lib.DeviceBuffer.from_array(np.ascontiguousarray(G, dtype=np.float64)), _lib.DeviceBuffer(n_s * 8)
    _lib.check(_lib.lib().srom_eigh_dev(dG.ptr, C.c_int64(n_s), dw.ptr, None), 'srom_eigh_dev')
    return dw.to_array((n_s,)), dG.to_array((n_s, n_s)).T


def modes_from_gramian(S_rows, G, tol, rom_dim=None, eigh=None):
    """Method of snapshots for a Gramian that is already on the host (e.g. after the all-reduce of the
    column-sharded build): eig(G) = Sigma^2 by the device eigensolver, W_k = leading eigenvectors / Sigma."""
    w, W = (eigh or _device_eigh)(G)
    w = np.maximum(w[::-1], 0.0)
    W = W[:, ::-1]
    Sigma = np.sqrt(w)
    k = energy_truncation(Sigma, tol) if rom_dim is None else int(rom_dim)
    Wk = np.ascontiguousarray(W[:, :k] / Sigma[:k])
    return Wk, k, Sigma


def compute_POD(snapshots, tol, rom_dim=None):
    """pod.py:181-200 with the same arguments (snapshots is n_f x n_s) and return order
    (U_full, U, nbModes, Sigma).  The reference takes a thin SVD; here the basis comes from the snapshot
    Gramian (f64 MFMA kernel) -- identical subspace and singular values (up to the sign of each mode and
    the eps*(sigma_0/sigma_i)^2 accuracy of the small singular values).  U_full is not formed (None): only
    the kept modes are recovered (U = S^T W Sigma^-1)."""
    S_rows = np.ascontiguousarray(np.asarray(snapshots, dtype=np.float64).T)
    n_s, n_f = S_rows.shape
    return _compute_POD_dev(_lib.DeviceBuffer.from_array(S_rows), n_s, n_f, tol, rom_dim)


def _compute_POD_dev(dS, n_s, n_f, tol, rom_dim=None):
    """compute_POD on a resident snapshot matrix (n_s x n_f, one snapshot per row)."""
    L = _lib.lib()
    # everything stays in HBM: Gramian (MFMA kernel) -> eigh (Jacobi kernels; rocSOLVER above 4096 snapshots) -> mode selection -> U = S^T W Sigma^-1;
    # only the n_s eigenvalues cross to the host for the energy truncation
    dG, dw = _lib.DeviceBuffer(n_s * n_s * 8), _lib.DeviceBuffer(n_s * 8)
    _lib.check(L.srom_gramian_dev(dS.ptr, C.c_int64(n_s), C.c_int64(n_f), C.c_int64(n_f), dG.ptr, None), 'srom_gramian_dev')
    _lib.check(L.srom_eigh_dev(dG.ptr, C.c_int64(n_s), dw.ptr, None), 'srom_eigh_dev')
    w = dw.to_array((n_s,))
    Sigma = np.sqrt(np.maximum(w[::-1], 0.0))
    k = energy_truncation(Sigma, tol) if rom_dim is None else int(rom_dim)
    dW, dU = _lib.DeviceBuffer(n_s * k * 8), _lib.DeviceBuffer(n_f * k * 8)
    _lib.check(L.srom_select_modes_dev(dG.ptr, dw.ptr, C.c_int64(n_s), C.c_int(k), dW.ptr, None), 'srom_select_modes_dev')
    _lib.check(L.srom_modes_dev(dS.ptr, C.c_int64(n_s), C.c_int64(n_f), C.c_int64(n_f), dW.ptr, C.c_int(k),
                                dU.ptr, None), 'srom_modes_dev')
    _lib.sync()
    U = dU.to_array((n_f, k))
    return None, U, k, Sigma[:min(n_s, n_f)]


def run_POD(snapshots_file, POD_file, config, rom_dim=None):
    """pod.py:110-141."""
    data = scutils.load_data(snapshots_file)
    snapshots = np.ascontiguousarray(get_snapshots(data, config.pod_type), dtype=np.float64)
    # one upload: preprocessing (pod.py:157-178) and the decomposition (pod.py:181-200) share the resident snapshot matrix
    wanted = [p_ for p_ in ('normalize', 'substract_mean', 'clustering') if p_ in config.preprocess]
    if 'clustering' in wanted and not config.preprocess_args.get('nbr_clusters', 0) > 0:
        print('Not using kmeans because nbr_clusters not specified in config.preprocess_args dictionary')
        wanted.remove('clustering')
    dS, n_s, n_f = _lib.DeviceBuffer.from_array(snapshots), snapshots.shape[0], snapshots.shape[1]
    if wanted:
        dS, n_s, n_f = _process_snapshots_dev(dS, n_s, n_f, wanted, config.preprocess_args)
    U_full, U, rom_dim, Sigma = _compute_POD_dev(dS, n_s, n_f, config.pod_tolerance)
    print('Computed POD with tolerance {}, resulting in {} dimensional system'.format(config.pod_tolerance, rom_dim))
    POD_info = {'U': U, 'q_ref': data['q'][0], 'v_ref': np.zeros(data['v'][0].shape)}
    results = {'POD_info': POD_info, 'config': vars(config), 'Sigma': Sigma}
    print('Saving POD data to {}'.format(POD_file))
    scutils.save_data(POD_file, results)
    return results
