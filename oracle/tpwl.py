"""Oracle: TPWL piecewise-affine reduced model (test infrastructure only).

Restates sofacontrol/tpwl/tpwl.py, sofacontrol/utils.py:302-335 (zoh) and
sofacontrol/scp/models/tpwl.py in numpy float64.  A model is a plain dict:
  q (P,r)  v (P,r)  u (P,m)  A_c (P,n,n)  B_c (P,n,m)  d_c (P,n)  w_q  w_v
with n = 2r and reduced state x = [v; q].
"""
import numpy as np
from scipy.linalg import expm


def nearest_point(model, x):
    """sofacontrol/tpwl/tpwl.py:160-168 -- argmin_i w_q||q_i-q|| + w_v||v_i-v|| (first minimum)."""
    r = model['q'].shape[1]
    v, q = x[:r], x[r:]
    qd = model['w_q'] * np.linalg.norm(model['q'] - q, axis=1)
    vd = model['w_v'] * np.linalg.norm(model['v'] - v, axis=1)
    return int(np.argmin(qd + vd))


def nearest_points(model, X):
    return np.array([nearest_point(model, x) for x in X], dtype=np.int32)


def weighting_factors(model, x, beta):
    """sofacontrol/tpwl/tpwl.py:170-191."""
    r = model['q'].shape[1]
    v, q = x[:r], x[r:]
    d = model['w_q'] * np.linalg.norm(model['q'] - q, axis=1) + \
        model['w_v'] * np.linalg.norm(model['v'] - v, axis=1)
    i = np.argmin(d)
    m = d[i]
    if m == 0:
        w = np.zeros_like(d)
        w[i] = 1
        return w
    w = np.exp(-beta * d / m)
    return w / np.sum(w)


def weighted_jacobians(model, x, beta, dt=None, method='zoh'):
    """sofacontrol/tpwl/tpwl.py:244-250: softmin-blended continuous tables, discretised if dt is given."""
    w = weighting_factors(model, x, beta)
    A = np.einsum('i,ijk->jk', w, model['A_c'])
    B = np.einsum('i,ijk->jk', w, model['B_c'])
    d = np.einsum('i,ij->j', w, model['d_c'])
    if dt is not None:
        A, B, d = discretize(A, B, d, dt, method)
    return A, B, d


def rollout_weighted(model, x0, u, beta, dt, method='zoh'):
    """sofacontrol/tpwl/tpwl.py:193-216 with the weighting branch of get_jacobians at every step."""
    x = np.zeros((u.shape[0] + 1, x0.shape[0]))
    x[0] = x0
    for i in range(u.shape[0]):
        A, B, d = weighted_jacobians(model, x[i], beta, dt, method)
        x[i + 1] = A @ x[i] + B @ u[i] + d
    return x


def zoh_affine(A, B, d, dt):
    """sofacontrol/utils.py:302-335 -- expm([[A, B, d], [0, 0, 0]] dt)."""
    n, m = B.shape
    M = np.zeros((n + m + 1, n + m + 1))
    M[:n, :n] = A
    M[:n, n:n + m] = B
    M[:n, n + m] = d
    Z = expm(M * dt)
    return Z[:n, :n], Z[:n, n:n + m], Z[:n, n + m]


def discretize(A, B, d, dt, method):
    """sofacontrol/tpwl/tpwl.py:272-297."""
    n = A.shape[0]
    I = np.eye(n)
    if method == 'fe':
        return I + dt * A, dt * B, dt * d
    if method == 'be':
        Ad = np.linalg.inv(I - dt * A)
        sep = np.linalg.inv(A) @ (Ad - I)
        return Ad, sep @ B, sep @ d
    if method == 'bil':
        Ad = (I + 0.5 * dt * A) @ np.linalg.inv(I - 0.5 * dt * A)
        sep = np.linalg.inv(A) @ (Ad - I)
        return Ad, sep @ B, sep @ d
    if method == 'zoh':
        return zoh_affine(A, B, d, dt)
    raise ValueError(method)


def pre_discretize(model, dt, method):
    """sofacontrol/tpwl/tpwl.py:299-322 -- returns stacked (A_d, B_d, d_d)."""
    out = [discretize(model['A_c'][i], model['B_c'][i], model['d_c'][i], dt, method)
           for i in range(model['q'].shape[0])]
    return (np.stack([o[0] for o in out]), np.stack([o[1] for o in out]),
            np.stack([o[2] for o in out]))


def rollout(model, Ad, Bd, dd, x0, u):
    """sofacontrol/tpwl/tpwl.py:193-216 with nn + prediscretised tables (226-234, 336-339)."""
    N = u.shape[0]
    x = np.zeros((N + 1, x0.shape[0]))
    x[0] = x0
    for k in range(N):
        i = nearest_point(model, x[k])
        x[k + 1] = Ad[i] @ x[k] + Bd[i] @ u[k] + dd[i]
    return x


def continuous_dynamics(model, x, u):
    """sofacontrol/scp/models/tpwl.py:32-50 -- f = A_i x + B_i u + d_i at the nearest point."""
    i = nearest_point(model, x)
    A, B, d = model['A_c'][i], model['B_c'][i], model['d_c'][i]
    return A @ x + B @ u + d, A, B


def characteristic_vals(model):
    """sofacontrol/scp/models/tpwl.py:66-84."""
    x = np.concatenate((model['v'], model['q']), axis=1)
    x_char = np.abs(x).max(axis=0)
    f = np.stack([continuous_dynamics(model, x[i], model['u'][i])[0] for i in range(x.shape[0])])
    return x_char, np.abs(f).max(axis=0)


def characteristic_dx(model, Ad, Bd, dd):
    """sofacontrol/tpwl/tpwl.py:324-334."""
    x = np.concatenate((model['v'], model['q']), axis=1)
    dx = np.zeros_like(x)
    for i in range(x.shape[0]):
        j = nearest_point(model, x[i])
        dx[i] = Ad[j] @ x[i] + Bd[j] @ model['u'][i] + dd[j] - x[i]
    return np.abs(dx).max(axis=0)


def synthetic_model(r, m, P, seed=0, w_q=1.0, w_v=0.0):
    """Synthetic TPWL tables as fixed in SURVEY.md section 8(d) (no TPWL model ships with the
    reference: tpwl_model_snapshots.pkl is git-ignored).  Rayleigh damping alpha=2.5,
    beta=0.01 as examples/hardware/model.py:14-15."""
    rng = np.random.default_rng(seed)
    n = 2 * r
    q = 3.0 * rng.standard_normal((P, r))
    v = 0.3 * rng.standard_normal((P, r))
    u = rng.uniform(0.0, 1500.0, (P, m))
    A_c = np.zeros((P, n, n))
    B_c = np.zeros((P, n, m))
    d_c = np.zeros((P, n))
    for i in range(P):
        S = rng.standard_normal((r, r))
        K = np.diag(rng.uniform(50.0, 500.0, r)) + 0.1 * 0.5 * (S + S.T)
        A_c[i, :r, :r] = -(2.5 * np.eye(r) + 0.01 * K)
        A_c[i, :r, r:] = -K
        A_c[i, r:, :r] = np.eye(r)
        B_c[i, :r, :] = 0.1 * rng.standard_normal((r, m))
        d_c[i, :r] = 0.01 * rng.standard_normal(r)
    return dict(q=q, v=v, u=u, A_c=A_c, B_c=B_c, d_c=d_c, w_q=w_q, w_v=w_v)


def synthetic_output_matrix(r, n_z=6, seed=100):
    """H = Hf V for a tip-node selector Hf (examples/diamond/diamond.py:269): rows 0..2 pick
    three rows of U in the velocity block, rows 3..5 the same rows in the position block."""
    rng = np.random.default_rng(seed)
    rows = rng.standard_normal((n_z // 2, r)) / np.sqrt(r) * 6.0
    H = np.zeros((n_z, 2 * r))
    H[:n_z // 2, :r] = rows
    H[n_z // 2:, r:] = rows
    return H
