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
    """{x : A x <= b} (sofacontrol/utils.py:364-398; the OSQP re-projection option is not part of the
    hot path and is not provided)."""

    def __init__(self, A, b, with_reproject=False):
        if with_reproject:
            raise NotImplementedError('with_reproject needs osqp and is outside the hot path')
        self.A = np.asarray(A, dtype=np.float64)
        self.b = np.asarray(b, dtype=np.float64)
        self.with_reproject = False

    def contains(self, x):
        return not (np.max(self.A @ x - self.b) > 0)

    def get_constraint_violation(self, x):
        return np.linalg.norm(np.maximum(self.A @ x - self.b, 0))


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
