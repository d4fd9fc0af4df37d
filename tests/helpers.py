"""Shared builders for the GPU tests: product-side objects from the seeded synthetic models."""
import numpy as np
import scipy.sparse as sp

from oracle import tpwl as otpwl


def small_rom(n_nodes, r, seed):
    """Same generator as tests/golden/make_golden.py:small_rom."""
    rng = np.random.default_rng(seed)
    n_f = 3 * n_nodes
    U, _ = np.linalg.qr(rng.standard_normal((n_f, r)))
    q_ref = rng.uniform(-108, 107, n_f)
    v_ref = 0.01 * rng.standard_normal(n_f)
    return U, q_ref, v_ref


def tip_selector(node, num_nodes):
    """linearModel(nodes=[node], num_nodes).C  (sofacontrol/measurement_models.py:87-103): rows [v; q]."""
    C = sp.lil_matrix((6, 6 * num_nodes))
    for a in range(3):
        C[a, 3 * node + a] = 1.0
        C[3 + a, 3 * num_nodes + 3 * node + a] = 1.0
    return C.tocsr()


def product_tpwl(model, U, q_ref, v_ref, Hf, discr='zoh', method='nn', beta=None, Cf=None):
    from sofacontrol_amd.tpwl.tpwl import TPWLATV
    data = dict(q=model['q'], v=model['v'], u=model['u'], A_c=model['A_c'], B_c=model['B_c'], d_c=model['d_c'],
                rom_info=dict(type='POD', U=U, q_ref=q_ref, v_ref=v_ref))
    params = dict(tpwl_method=method, dist_weights={'q': model['w_q'], 'v': model['w_v']}, beta_weighting=beta)
    return TPWLATV(data=data, params=params, Hf=Hf, Cf=Cf, discr_method=discr)


def golden_problem(r, m, P, n_nodes, seed, q_scale=1.0):
    """Mirror of make_golden.make_problem."""
    model = otpwl.synthetic_model(r, m, P, seed=seed)
    model['q'] = model['q'] * q_scale
    U, q_ref, v_ref = small_rom(n_nodes, r, seed + 1)
    Hf = tip_selector(n_nodes // 2, n_nodes)
    return model, U, q_ref, v_ref, Hf


def Poly(A, b):
    """The product's Polyhedron (same protocol as sofacontrol/utils.py:364-398)."""
    from sofacontrol_amd.utils import Polyhedron
    return Polyhedron(A, b)


def meas_selector(nodes, num_nodes):
    """Position rows of the listed nodes out of x = [v; q] (mirror of make_golden.meas_selector)."""
    import scipy.sparse as sp
    n_f = 3 * num_nodes
    Cf = sp.lil_matrix((3 * len(nodes), 2 * n_f))
    for i, nd in enumerate(nodes):
        for a in range(3):
            Cf[3 * i + a, n_f + 3 * nd + a] = 1.0
    return Cf.tocsr()


def assembly_points(n_f, m, q_ref, seed, count=3):
    """Seeded synthetic full-order points (K, D, M, S, H, b, f, q, v, q+, v+, u) shared by the golden generator
    (g12_assembly) and the parity test, so that only the outputs are stored."""
    rng = np.random.default_rng(seed)
    pts = []
    for i in range(count):
        G = rng.standard_normal((n_f, n_f))
        K = G @ G.T / n_f + 5 * np.eye(n_f)
        D = 0.01 * K + 0.5 * np.eye(n_f)
        M = np.diag(rng.uniform(0.5, 2.0, n_f))
        S = M + 0.01 * D + 1e-4 * K
        H = rng.standard_normal((n_f, m))
        b = rng.standard_normal(n_f); f = rng.standard_normal(n_f)
        q = q_ref + rng.standard_normal(n_f); v = 0.1 * rng.standard_normal(n_f)
        pts.append(dict(K=K, D=D, M=M, S=S, H=H, b=b, f=f, q=q, v=v, q_next=q + 0.01 * v,
                        v_next=v + 0.01 * rng.standard_normal(n_f), u=rng.uniform(0, 100, m), t=0.01 * i, dt=0.01))
    return pts
