"""Oracle: POD reduced-order map (test infrastructure only, see oracle/__init__.py).

Restates sofacontrol/mor/pod.py in plain numpy float64.
"""
import numpy as np


def qv2x(q, v):
    """sofacontrol/utils.py:135-136 -- full/reduced state is x = [v; q]."""
    return np.concatenate((v, q), axis=-1)


def x2qv(x):
    """sofacontrol/utils.py:139-148 -- returns (q, v)."""
    n = x.shape[-1] // 2
    return x[..., n:], x[..., :n]


def make_V(U):
    """sofacontrol/mor/pod.py:19 -- V = kron(I2, U)."""
    return np.kron(np.eye(2), U)


def project(U, ref, Xf):
    """sofacontrol/mor/pod.py:46-50 -- U^T (qf - q_ref), rows of Xf are snapshots.

    Xf (B, n_f) or (n_f,) -> (B, r) or (r,).
    """
    return (Xf - ref) @ U


def project_x(U, q_ref, v_ref, Xf):
    """sofacontrol/mor/pod.py:51-52 -- V^T (xf - x_ref) with x = [v; q]."""
    n_f = U.shape[0]
    v, q = Xf[..., :n_f], Xf[..., n_f:]
    return np.concatenate(((v - v_ref) @ U, (q - q_ref) @ U), axis=-1)


def lift(U, ref, Xr):
    """sofacontrol/mor/pod.py:30-33 -- U q + q_ref (rows of Xr are reduced vectors)."""
    return Xr @ U.T + ref


def lift_x(U, q_ref, v_ref, Xr):
    """sofacontrol/mor/pod.py:34-35 -- V x + x_ref."""
    r = U.shape[1]
    return np.concatenate((Xr[..., :r] @ U.T + v_ref, Xr[..., r:] @ U.T + q_ref), axis=-1)


def reduce_matrix(U, M, left=False, right=False):
    """sofacontrol/mor/pod.py:56-72 -- U^T M U / U^T M / M U."""
    if (left and right) or (not left and not right):
        return U.T @ M @ U
    if left:
        return U.T @ M
    return M @ U


def get_snapshots(data, pod_type):
    """sofacontrol/mor/pod.py:144-154."""
    if pod_type == 'q':
        return np.asarray(data['q']) - data['q'][0]
    if pod_type == 'v':
        return np.asarray(data['v'])
    if pod_type == 'a':
        return np.asarray(data['v+']) - np.asarray(data['v'])
    raise ValueError(pod_type)


def process_snapshots(snapshots, preprocess, args):
    """sofacontrol/mor/pod.py:157-178 (snapshots: n_s x n_f, one per row)."""
    snapshots = np.array(snapshots, dtype=float)
    if 'normalize' in preprocess:
        snapshots = (snapshots - snapshots.min(axis=0)) / (snapshots.max(axis=0) + 1e-15 - snapshots.min(axis=0))
    if 'substract_mean' in preprocess:
        snapshots = snapshots - snapshots.mean(axis=0, keepdims=True)
    if 'clustering' in preprocess and args.get('nbr_clusters', 0) > 0:
        snapshots = kmeans_centroids(snapshots, args['nbr_clusters'])
    return snapshots


def kmeans_centroids(X, k, n_init=100, max_iter=1000, random_state=0, tol=1e-4):
    """sofacontrol/mor/pod.py:207-216 calls sklearn.cluster.KMeans(k, n_init=100, max_iter=1000, random_state=0) -- a
    third-party dependency of the reference (scikit-learn, unpinned there; 1.7.2 in this image).  This is a plain-numpy
    restatement of that estimator's published dense algorithm (`sklearn/cluster/_kmeans.py`: centring, k-means++ seeding
    with 2 + int(log k) local trials, Lloyd iterations, empty-cluster relocation, best of n_init by inertia), pinned by
    tests/golden g19 (the imported reference's own output) and, where sklearn is installed, against sklearn itself.  Not
    reproduced: the estimator's tie-breaking when k exceeds the number of DISTINCT snapshots (all distances zero; it warns
    about that case itself)."""
    X = np.array(X, dtype=float)
    n_s, n_f = X.shape
    mean = X.mean(axis=0)
    X = X - mean
    xn = (X * X).sum(axis=1)
    tol_abs = np.mean(np.var(X, axis=0)) * tol
    rs = np.random.RandomState(random_state)
    n_trials = 2 + int(np.log(k))

    def sqdist(Y):
        return np.maximum((Y * Y).sum(axis=1)[:, None] - 2.0 * Y @ X.T + xn[None, :], 0.0)

    def labels_of(Cc):
        return np.argmin((Cc * Cc).sum(axis=1)[None, :] - 2.0 * X @ Cc.T, axis=1)

    best = None
    for _ in range(n_init):
        idx = np.empty(k, dtype=int)
        idx[0] = rs.choice(n_s, p=np.full(n_s, 1.0 / n_s))
        closest = sqdist(X[idx[:1]])[0]
        pot = closest.sum()
        for c in range(1, k):
            cand = np.searchsorted(np.cumsum(closest), rs.uniform(size=n_trials) * pot)
            np.clip(cand, None, n_s - 1, out=cand)
            d = np.minimum(closest, sqdist(X[cand]))
            pots = d.sum(axis=1)
            b = int(np.argmin(pots))
            pot, closest, idx[c] = pots[b], d[b], cand[b]
        Cc = X[idx].copy()
        labels_old = np.full(n_s, -1)
        strict = False
        for _it in range(max_iter):
            labels = labels_of(Cc)
            sums = np.zeros_like(Cc)
            np.add.at(sums, labels, X)
            counts = np.bincount(labels, minlength=k).astype(float)
            empty = np.where(counts == 0)[0]
            if len(empty):
                dist = ((X - Cc[labels]) ** 2).sum(axis=1)
                far = np.argpartition(dist, -len(empty))[:-len(empty) - 1:-1]      # the estimator's own selection (ties included)
                for e, f in zip(empty, far):
                    sums[labels[f]] -= X[f]; sums[e] = X[f]; counts[e] = 1; counts[labels[f]] -= 1
            Cn = np.where(counts[:, None] > 0, sums / np.maximum(counts, 1)[:, None], Cc)
            shift = ((Cn - Cc) ** 2).sum()
            Cc = Cn
            if np.array_equal(labels, labels_old):
                strict = True
                break
            if shift <= tol_abs:
                break
            labels_old = labels
        if not strict:
            labels = labels_of(Cc)
        inertia = ((X - Cc[labels]) ** 2).sum()
        if best is None or (inertia < best[0] and not _same_clustering(labels, best[1], k)):
            best = (inertia, labels, Cc)
    return best[2] + mean


def _same_clustering(a, b, k):
    m = np.full(k, -1)
    for la, lb in zip(a, b):
        if m[la] == -1:
            m[la] = lb
        elif m[la] != lb:
            return False
    return True


def energy_truncation(S, tol):
    """sofacontrol/mor/pod.py:192-197 -- smallest k>=1 with sum(S[k:]^2)/sum(S^2) <= tol."""
    s2 = S ** 2
    i = 0
    while (np.sum(s2[i:]) / np.sum(s2)) > tol or i == 0:
        i += 1
    return i


def compute_pod(snapshots, tol):
    """sofacontrol/mor/pod.py:181-200 -- thin SVD of (n_f x n_s) + energy truncation."""
    U_full, S, _ = np.linalg.svd(snapshots, full_matrices=False)
    k = energy_truncation(S, tol)
    return U_full, U_full[:, :k], k, S


def gramian(S_rows):
    """Snapshot Gramian G = S S^T for S (n_s x n_f) row-major snapshots.

    The reference takes the SVD of S^T (pod.py:190); the singular values are the square
    roots of eig(G) and the left singular vectors of S^T are S^T W Sigma^-1.
    """
    return S_rows @ S_rows.T


def pod_from_gramian(S_rows, tol):
    """POD basis through the method of snapshots; same outputs as compute_pod(S_rows.T, tol)
    up to the sign of each mode."""
    G = gramian(S_rows)
    w, W = np.linalg.eigh(G)
    order = np.argsort(w)[::-1]
    w = np.maximum(w[order], 0.0)
    W = W[:, order]
    S = np.sqrt(w)
    k = energy_truncation(S, tol)
    U = S_rows.T @ (W[:, :k] / S[:k])
    return U, k, S
