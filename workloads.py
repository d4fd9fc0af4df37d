"""Synthetic workloads of the BASELINE.json configurations (pure numpy; shared by bench.py, smoke and tests).

No TPWL model ships with the reference (tpwl_model_snapshots.pkl is git-ignored) and no data set
travels to the GPU box, so every input is generated from seeds.  Shapes, cost weights, constraints and
the target follow the reference's Diamond / Trunk drivers (cited inline; paths relative to the
reference root); magnitudes of the synthetic reduced dynamics are chosen so that the figure-8 target is
reachable inside the cable limits and the horizon crosses several TPWL regions.
"""
import numpy as np
from scipy.linalg import expm


def pod_basis(n_f=4884, r=30, seed=0):
    """Orthonormal basis U (n_f x r) and reference configuration (range of the shipped q_ref)."""
    U, _ = np.linalg.qr(np.random.default_rng(seed).standard_normal((n_f, r)))
    q_ref = np.random.default_rng(seed + 1).uniform(-108.0, 107.0, n_f)
    v_ref = np.zeros(n_f)
    return np.ascontiguousarray(U), q_ref, v_ref


def snapshots(q_ref, B, seed=2, chunk=4096):
    """B full-order position snapshots (B x n_f): q_ref + 5 N(0,1)."""
    rng = np.random.default_rng(seed)
    X = np.empty((B, q_ref.shape[0]))
    for i in range(0, B, chunk):
        j = min(B, i + chunk)
        X[i:j] = q_ref + 5.0 * rng.standard_normal((j - i, q_ref.shape[0]))
    return X


def tip_selector_rows(U, node):
    """H = Hf V for Hf = linearModel(nodes=[node]).C (examples/diamond/diamond.py:269): rows
    [v_x v_y v_z q_x q_y q_z] of the tip node."""
    r = U.shape[1]
    rows = U[3 * node:3 * node + 3, :]
    H = np.zeros((6, 2 * r))
    H[:3, :r] = rows
    H[3:, r:] = rows
    return H


def tpwl_tables(r, m, P, seed=10, q_spread=300.0, v_spread=300.0, b_scale=40.0, u_max=1500.0, k_var=0.005,
                b_var=0.003, d_scale=0.01, s_scale=0.05):
    """P linearisation points of a lightly damped second-order reduced model
    A_c = [[-(alpha I + beta K_i), -K_i], [I, 0]] with Rayleigh damping alpha = 2.5, beta = 0.01
    (examples/hardware/model.py:14-15), K_i = diag(U(50,500)) + symmetric perturbation."""
    rng = np.random.default_rng(seed)
    n = 2 * r
    q = q_spread * rng.standard_normal((P, r)) / np.sqrt(r)
    q[0] = 0.0                                   # the rest configuration is a model point
    v = v_spread * rng.standard_normal((P, r)) / np.sqrt(r)
    u = rng.uniform(0.0, u_max, (P, m))
    A_c = np.zeros((P, n, n)); B_c = np.zeros((P, n, m)); d_c = np.zeros((P, n))
    K0 = np.diag(rng.uniform(50.0, 500.0, r))
    B0 = b_scale * rng.standard_normal((r, m))
    for i in range(P):
        S = rng.standard_normal((r, r))
        K = K0 * (1.0 + k_var * rng.standard_normal()) + s_scale * 0.5 * (S + S.T)
        A_c[i, :r, :r] = -(2.5 * np.eye(r) + 0.01 * K)
        A_c[i, :r, r:] = -K
        A_c[i, r:, :r] = np.eye(r)
        B_c[i, :r, :] = B0 * (1.0 + b_var * rng.standard_normal((r, m)))
        d_c[i, :r] = d_scale * rng.standard_normal(r)
    d_c[0] = 0.0
    return dict(q=q, v=v, u=u, A_c=A_c, B_c=B_c, d_c=d_c)


def zoh_tables(tab, dt):
    """Zero-order-hold discretisation of every point (sofacontrol/utils.py:302-335)."""
    P, n, m = tab['B_c'].shape
    Ad = np.empty((P, n, n)); Bd = np.empty((P, n, m)); dd = np.empty((P, n))
    M = np.zeros((n + m + 1, n + m + 1))
    for i in range(P):
        M[:n, :n] = tab['A_c'][i]; M[:n, n:n + m] = tab['B_c'][i]; M[:n, n + m] = tab['d_c'][i]
        Z = expm(M * dt)
        Ad[i], Bd[i], dd[i] = Z[:n, :n], Z[:n, n:n + m], Z[:n, n + m]
    return Ad, Bd, dd


def figure8(z_ref, T=10.0, M=1000, scale=1.0):
    """Figure-8 of the tip (examples/diamond/diamond.py:277-283), returned relative to z_ref."""
    t = np.linspace(0, T, M)
    th = np.linspace(0, 2 * np.pi, M)
    zf = np.zeros((M, 6))
    zf[:, 3] = scale * (-20.0 * np.sin(th)) - 5.5
    zf[:, 4] = scale * (10.0 * np.sin(2 * th)) + 1.5
    z = zf.copy()
    z[:, 3] -= -5.5      # the synthetic rest tip position is taken as (-5.5, 1.5): target relative to it
    z[:, 4] -= 1.5
    return t, z


def trunk_c5(r=30, N=50, dt=0.1, seed=20, P=64, n_f=2127, tip_node=51):
    """BASELINE config C5: Trunk n_f = 2127 (3 x 709), POD r = 30 (n_x = 60), n_u = 8, SCP horizon N = 50, dt = 0.1
    (examples/trunk/trunk.py:292-316, dt: line 309): Qz = diag(0,0,0,100,100,0), R = 1e-5 I, U = [0,800]^8, X = None.
    (The target is the Diamond's figure-8 of `figure8`, twice the Trunk driver's x amplitude.)"""
    m = 8
    U, q_ref, v_ref = pod_basis(n_f, r, seed=3)
    H = tip_selector_rows(U, tip_node)
    tab = tpwl_tables(r, m, P, seed=seed, u_max=800.0)
    Ad, Bd, dd = zoh_tables(tab, dt)
    Qz = np.zeros((6, 6)); Qz[3, 3] = 100.0; Qz[4, 4] = 100.0
    R = 1e-5 * np.eye(m)
    UA = np.kron(np.eye(m), np.array([[1.0], [-1.0]]))
    Ub = np.tile([800.0, 0.0], m)
    t, z = figure8(None)
    return dict(U=U, q_ref=q_ref, v_ref=v_ref, H=H, tab=tab, Ad=Ad, Bd=Bd, dd=dd, dt=dt, N=N, Qz=Qz, R=R,
                UA=UA, Ub=Ub, XA=None, Xb=None, t=t, z=z, r=r, m=m, P=P, tip_node=tip_node)


def diamond_c2(r=30, N=50, dt=0.05, seed=10, P=64, n_f=4884, tip_node=1354, with_X=True):
    """BASELINE config C2: Diamond n_f = 4884, POD r = 30 (n_x = 60), n_u = 4, SCP horizon N = 50,
    dt = 0.05 (examples/diamond/diamond.py:285-310): Qz = diag(0,0,0,100,100,0), R = 1e-5 I,
    U = [0,1500]^4, X = box on the tip x/y (diamond.py:295-304)."""
    m = 4
    U, q_ref, v_ref = pod_basis(n_f, r, seed=0)
    H = tip_selector_rows(U, tip_node)
    # scale the basis-row output map so that reduced coordinates of O(100) move the tip by O(20 mm)
    tab = tpwl_tables(r, m, P, seed=seed)
    Ad, Bd, dd = zoh_tables(tab, dt)
    Qz = np.zeros((6, 6)); Qz[3, 3] = 100.0; Qz[4, 4] = 100.0
    R = 1e-5 * np.eye(m)
    UA = np.kron(np.eye(m), np.array([[1.0], [-1.0]]))
    Ub = np.tile([1500.0, 0.0], m)
    Hz = np.zeros((2, 6)); Hz[0, 3] = 1; Hz[1, 4] = 1
    Hx = Hz @ H
    XA = np.vstack([-Hx, Hx])
    Xb = np.array([17.5, 20.0, 17.5, 20.0])       # |x| <= 17.5, |y| <= 20 around the rest tip
    t, z = figure8(None)
    return dict(U=U, q_ref=q_ref, v_ref=v_ref, H=H, tab=tab, Ad=Ad, Bd=Bd, dd=dd, dt=dt, N=N, Qz=Qz, R=R,
                UA=UA, Ub=Ub, XA=XA if with_X else None, Xb=Xb if with_X else None, t=t, z=z, r=r, m=m, P=P)


def ssm_model(n, m, rom_order, ssm_order, seed=0):
    """Seeded SSM polynomial model (SSM/ssm.py shapes): damped oscillator pairs in the linear part, small random
    higher-order terms.  Returns the coefficient arrays of `SSM.__init__` (r_coeff, B, w_coeff, v_coeff, rd_coeff,
    Bd) and z_ref; monomial counts comb(dim + order, order) - 1."""
    from math import comb
    rng = np.random.default_rng(seed)
    nr, ns = comb(n + rom_order, rom_order) - 1, comb(n + ssm_order, ssm_order) - 1
    R = np.zeros((n, nr))
    for k in range(n // 2):
        w, zt = 3.0 + 2.0 * k, 0.5 + 0.3 * k
        R[2 * k:2 * k + 2, 2 * k:2 * k + 2] = [[-zt, -w], [w, -zt]]
    if n % 2:
        R[n - 1, n - 1] = -1.0
    R[:, n:] = 0.2 * rng.standard_normal((n, nr - n))
    W = np.zeros((n, ns)); W[:, :n] = np.eye(n) + 0.1 * rng.standard_normal((n, n))
    W[:, n:] = 0.1 * rng.standard_normal((n, ns - n))
    V = np.zeros((n, ns)); V[:, :n] = np.linalg.inv(W[:, :n])
    V[:, n:] = 0.1 * rng.standard_normal((n, ns - n))
    B = rng.standard_normal((n, m))
    Rd = np.zeros((n, nr)); Rd[:, :n] = np.eye(n)
    Rd = Rd + 0.01 * R
    Bd = 0.01 * B
    z_ref = rng.standard_normal(n)
    return dict(n=n, m=m, rom_order=rom_order, ssm_order=ssm_order, R=R, B=B, W=W, V=V, Rd=Rd, Bd=Bd, z_ref=z_ref)


def ssm_c3(problems=256, rank=0):
    """BASELINE config C3 as bench.py times it and tests/test_ssm_gpu.py checks it: SSM reduction r = 10 (n_x = 10, cubic
    reduced dynamics, quadratic maps: 285 / 65 monomials), n_u = 8, iLQR horizon N = 100 at dt = 0.05 with the backward-Euler
    discretisation (examples/trunk/trunk.py:365), `problems` independent tracking problems (circle-like targets of growing
    amplitude on the first two outputs).  Returns the seeded model and the problem arrays; the first k problems of a
    larger batch are the k-problem batch (the parity test solves a prefix of what the bench solves)."""
    n, m, N, dt = 10, 8, 100, 0.05
    model = ssm_model(n, m, 3, 2, seed=95)
    Qz = np.diag([100.] * 3 + [1.] * 7)
    R = np.eye(m)
    rng = np.random.default_rng(2 + rank)
    x0 = 0.05 * rng.standard_normal((problems, n))
    th = np.linspace(0, 2 * np.pi, N + 1)
    zt = np.zeros((problems, N + 1, n))
    zt[:, :, 0] = 0.1 * np.sin(th)[None, :] * (1 + np.arange(problems)[:, None] / 256.0)
    zt[:, :, 1] = 0.1 * (1 - np.cos(th))[None, :]
    zt = zt + model['z_ref']
    return dict(n=n, m=m, N=N, dt=dt, discr='be', model=model, Qz=Qz, R=R, Qf=Qz, x0=x0, zt=zt)
