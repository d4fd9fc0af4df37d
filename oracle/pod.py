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
