"""Measurement / output models y = C x on the full-order state x = [v; q] (sofacontrol/measurement_models.py:7-107):
sparse node selectors -- the `Cf` / `Hf` the drivers hand to the TPWL model (examples/diamond/diamond.py:268-271).
Host-side data format only (a 6-row gather per node); the reduced matrices C = Cf V, H = Hf V are formed once in
TPWL.set_measurement_model / set_output_model."""
import numpy as np
from scipy.linalg import block_diag
from scipy.sparse import coo_matrix, vstack

from .utils import x2qv


def _selector(nodes, num_nodes, offset):
    nodes = np.asarray(nodes, dtype=np.int64)
    rows = np.arange(3 * nodes.size)
    cols = offset + 3 * np.repeat(nodes, 3) + np.tile(np.arange(3), nodes.size)
    return coo_matrix((np.ones(rows.size), (rows, cols)), shape=(3 * nodes.size, 6 * num_nodes)).tolil()


def buildCq(nodes, num_nodes):
    """Position rows: x = [v_0 .. v_{N-1}; q_0 .. q_{N-1}], three components per node (measurement_models.py:87-94)."""
    return _selector(nodes, num_nodes, 3 * num_nodes)


def buildCv(nodes, num_nodes):
    """Velocity rows (measurement_models.py:97-103)."""
    return _selector(nodes, num_nodes, 0)


class linearModel:
    """measurement_models.py:7-44: velocity rows first, then position rows."""

    def __init__(self, nodes, num_nodes, pos=True, vel=True, qv=False):
        self.pos = pos
        self.vel = vel
        self.build_C_matrix(nodes, num_nodes)
        self.num_nodes = num_nodes

    def build_C_matrix(self, nodes, num_nodes):
        if self.vel and not self.pos:
            self.C = buildCv(nodes, num_nodes)
        elif self.pos and not self.vel:
            self.C = buildCq(nodes, num_nodes)
        else:
            self.C = vstack((buildCv(nodes, num_nodes), buildCq(nodes, num_nodes)))

    def evaluate(self, x, qv=False):
        z = self.C @ x
        return np.concatenate(x2qv(z)) if qv else z


class MeasurementModel(linearModel):
    """measurement_models.py:47-84: the same selector with additive Gaussian noise N(mean, covariance)."""

    def __init__(self, nodes, num_nodes, pos=True, vel=True, mu_q=None, S_q=None, mu_v=None, S_v=None, qv=False):
        super().__init__(nodes, num_nodes, pos=pos, vel=vel)
        rows = self.C.shape[0]
        pos_dim = rows // 2 if (pos and vel) else (rows if pos else 0)
        vel_dim = rows // 2 if (pos and vel) else (rows if vel else 0)
        mu_q = np.zeros(pos_dim) if mu_q is None else mu_q
        mu_v = np.zeros(vel_dim) if mu_v is None else mu_v
        S_q = np.zeros((pos_dim, pos_dim)) if S_q is None else S_q
        S_v = np.zeros((vel_dim, vel_dim)) if S_v is None else S_v
        self.mean = np.concatenate((mu_v, mu_q))
        self.covariance = block_diag(S_v, S_q)
        self.qv = qv
        assert self.mean.shape[0] == rows
        assert self.covariance.shape == (rows, rows)

    def evaluate(self, x):
        z = self.C @ x + np.random.multivariate_normal(mean=self.mean, cov=self.covariance)
        return np.concatenate(x2qv(z)) if self.qv else z
