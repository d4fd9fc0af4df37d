"""Oracle: SSM polynomial reduced model (test infrastructure only).

Restates sofacontrol/SSM/ssm.py: get_poly_basis 158-164, maps 167-178, Jacobians 198-235,
discretize_dynamics 279-301, update_dynamics 331-333, rollout 134-156, compute_RO_state 338-344.
The reference differentiates the lambdified maps with jax (absent from the build container, and float32
by default there); the derivatives here are the analytic monomial derivatives.  Pinned by
tests/golden/g10_ssm.npz: the imported reference run with numpy in place of jax.numpy and complex-step
differentiation in place of jax.jacobian (tests/golden/_ref_import.py) -- parity against jax itself is
unpinned."""
import numpy as np


def exponents(dim, order):
    """Exponent table (n_mon, dim) of sorted(itermonomials(zeta, order), key=monomial_key('grevlex',
    reversed(zeta)))[1:]: graded, and within a degree lexicographic with x1 first (ssm.py:158-164)."""
    out = []

    def rec(pos, left, cur):
        if pos == dim - 1:
            out.append(cur + [left])
            return
        for e in range(left, -1, -1):
            rec(pos + 1, left - e, cur + [e])
    for deg in range(1, order + 1):
        rec(0, deg, [])
    return np.array(out, dtype=np.int64)


def phi(E, x):
    return np.prod(np.power(x[None, :], E), axis=1)


def dphi(E, x):
    """(n_mon, dim): d phi_j / d x_i."""
    n_mon, dim = E.shape
    D = np.zeros((n_mon, dim))
    for i in range(dim):
        Ei = E.copy()
        Ei[:, i] = np.maximum(Ei[:, i] - 1, 0)
        D[:, i] = E[:, i] * np.prod(np.power(x[None, :], Ei), axis=1)
    return D


def make_model(n, m, rom_order, ssm_order, r_coeff, B, w_coeff, v_coeff, z_ref, rd_coeff=None, Bd=None):
    return dict(n=n, m=m, Er=exponents(n, rom_order), Es=exponents(n, ssm_order), R=r_coeff, B=B, W=w_coeff,
                V=v_coeff, z_ref=z_ref, Rd=rd_coeff, Bd=Bd)


def dynamics(model, x, u, discrete=False):
    """ssm.py:167-168 / 177-178."""
    R, B = (model['Rd'], model['Bd']) if discrete else (model['R'], model['B'])
    return R @ phi(model['Er'], x) + B @ u


def continuous_jacobians(model, x, u, discrete=False):
    """ssm.py:198-212: A, B = d f / d(x, u); d = f - A x - B u."""
    R, B = (model['Rd'], model['Bd']) if discrete else (model['R'], model['B'])
    A = R @ dphi(model['Er'], x)
    d = dynamics(model, x, u, discrete) - A @ x - B @ u
    return A, B.copy(), d


def discretize(A_c, B_c, d_c, dt, method):
    """ssm.py:279-301 (zoh is not offered by the reference's SSM class)."""
    I = np.eye(A_c.shape[0])
    if method == 'fe':
        return I + dt * A_c, dt * B_c, dt * d_c
    if method == 'be':
        A_d = np.linalg.inv(I - dt * A_c)
    elif method == 'bil':
        A_d = (I + 0.5 * dt * A_c) @ np.linalg.inv(I - 0.5 * dt * A_c)
    else:
        raise RuntimeError('self.discr_method must be in [fe, be, bil, zoh]')
    sep = np.linalg.inv(A_c) @ (A_d - I)
    return A_d, sep @ B_c, sep @ d_c


def jacobians(model, x, u, dt, method='fe', discrete=False):
    """ssm.py:214-218."""
    if discrete:
        return continuous_jacobians(model, x, u, discrete=True)
    A, B, d = continuous_jacobians(model, x, u)
    return discretize(A, B, d, dt, method)


def observe(model, x):
    """C_map, ssm.py:170-171 (without z_ref)."""
    return model['W'] @ phi(model['Es'], x)


def observer_jacobians(model, x):
    """ssm.py:220-227."""
    H = model['W'] @ dphi(model['Es'], x)
    return H, observe(model, x) - H @ x


def reduce(model, z):
    """compute_RO_state, ssm.py:338-344."""
    return model['V'] @ phi(model['Es'], z - model['z_ref'])


def rollout(model, x0, u, dt, method='fe', discrete=False):
    """ssm.py:134-156."""
    N = u.shape[0]
    x = np.zeros((N + 1, x0.shape[0]))
    x[0] = x0
    for i in range(N):
        A, B, d = jacobians(model, x[i], u[i], dt, method, discrete)
        x[i + 1] = A @ x[i] + B @ u[i] + d
    z = np.stack([observe(model, xi) for xi in x]) + model['z_ref']
    return x, z


def synthetic(n, m, rom_order, ssm_order, seed=0):
    """Seeded SSM model (the generator lives in workloads.py, shared with bench.py) as an oracle model dict."""
    import workloads
    d = workloads.ssm_model(n, m, rom_order, ssm_order, seed)
    return make_model(n, m, rom_order, ssm_order, d['R'], d['B'], d['W'], d['V'], d['z_ref'], d['Rd'], d['Bd'])
