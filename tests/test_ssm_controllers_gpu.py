"""GPU parity: SSM closed-loop controller (sofacontrol/SSM/controllers.py:16-310) against the golden sequence g15 of the
imported reference, the in-process asynchronous solver client on an SSM GuSTO node, and the measurement re-projection
(utils.py:364-407) against the exact projection."""
import io
import contextlib
import itertools

import numpy as np
import pytest

from oracle import ssm as ossm
from test_controllers_gpu import FakeGuSTOClient
from test_ssm_gpu import product_ssm

pytestmark = pytest.mark.gpu


def quiet(f, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return f(*a, **k)


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_ssm_scp_controller_golden(golden, tag):
    import sofacontrol_amd.SSM.controllers as sctl
    g = golden('g15_ssm_controllers')
    model = ossm.synthetic(6, 4, 3, 3, seed=150)
    s = product_ssm(model)
    N_replan, delay, steps, dt = g[tag + '_params']
    c = sctl.scp(s, None, float(dt), N_replan=int(N_replan), delay=float(delay), client=FakeGuSTOClient())
    c.set_sim_timestep(float(dt))
    us, xs = [], []
    for k in range(int(steps)):
        us.append(quiet(c.evaluate, k * dt, g[tag + '_y'][k], None, np.zeros(4)))
        xs.append(c.observer.x.copy())
    np.testing.assert_allclose(np.stack(xs), g[tag + '_x_obs'], rtol=0, atol=1e-11 * max(1.0, np.abs(g[tag + '_x_obs']).max()))
    np.testing.assert_allclose(np.stack(us), g[tag + '_u'], rtol=0, atol=1e-9 * max(1.0, np.abs(g[tag + '_u']).max()))
    np.testing.assert_allclose(c.observer.z, g[tag + '_z_obs'], rtol=0, atol=1e-13)
    info = c.save_controller_info()
    assert sorted(info) == ['rollout_time', 'solve_times', 't_opt', 't_rollout', 'u_opt', 'z_opt', 'z_rollout']
    np.testing.assert_allclose(info['t_opt'], g[tag + '_t_opt'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(info['u_opt'], g[tag + '_u_opt'], rtol=0, atol=1e-9 * np.abs(g[tag + '_u_opt']).max())
    np.testing.assert_allclose(info['z_opt'], g[tag + '_z_opt'], rtol=0, atol=1e-9 * max(1.0, np.abs(g[tag + '_z_opt']).max()))
    np.testing.assert_allclose(info['z_rollout'][0], g[tag + '_z_rollout0'], rtol=0, atol=1e-9 * max(1.0, np.abs(g[tag + '_z_rollout0']).max()))
    np.testing.assert_allclose(info['t_rollout'][0], g[tag + '_t_rollout0'], rtol=0, atol=1e-12)
    assert len(info['solve_times']) == int(g[tag + '_n_solves']) and info['rollout_time'] == float(g[tag + '_rollout_time'])


def _exact_projection(A, b, x):
    """Brute force over active sets (small polyhedra): the KKT point with the smallest distance."""
    best, bestd = None, np.inf
    mc, n = A.shape
    for k in range(0, min(mc, n) + 1):
        for act in itertools.combinations(range(mc), k):
            if k == 0:
                p = x.copy()
            else:
                Aa = A[list(act)]
                try:
                    lam = np.linalg.solve(Aa @ Aa.T, Aa @ x - b[list(act)])
                except np.linalg.LinAlgError:
                    continue
                if (lam < -1e-12).any():
                    continue
                p = x - Aa.T @ lam
            if (A @ p - b).max() <= 1e-10:
                d = np.linalg.norm(p - x)
                if d < bestd:
                    best, bestd = p, d
    return best


@pytest.mark.parametrize('n,extra', [(2, 1), (3, 2), (6, 0), (6, 3)])
def test_polyhedron_projection_exact(n, extra):
    """Polyhedron(with_reproject=True).project_to_polyhedron: box + `extra` random facets; points inside come back
    unchanged, points outside land on the exact Euclidean projection (the reference's OSQP stops at eps = 1e-3)."""
    from sofacontrol_amd.utils import Polyhedron
    rng = np.random.default_rng(10 * n + extra)
    A = np.vstack([np.kron(np.eye(n), np.array([[1.], [-1.]]))] + ([rng.standard_normal((extra, n))] if extra else []))
    b = np.concatenate([np.tile([1.0, 0.5], n), 0.8 + rng.uniform(0, 0.5, extra)])
    P = Polyhedron(A, b, with_reproject=True)
    for trial in range(6):
        x = rng.uniform(-2.5, 2.5, n) * (1.0 if trial else 0.1)
        got = P.project_to_polyhedron(x)
        if P.contains(x):
            np.testing.assert_array_equal(got, x)
            continue
        want = _exact_projection(A, b, x)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
        assert (A @ got - b).max() <= 1e-9
    X = rng.uniform(-2.5, 2.5, (5, n))
    got = P.project_to_polyhedron(X)
    for i in range(5):
        np.testing.assert_allclose(got[i], _exact_projection(A, b, X[i]), rtol=0, atol=1e-9)
    with pytest.raises(RuntimeError):
        Polyhedron(A, b).project_to_polyhedron(X[0])


def test_ssm_controller_reprojects_measurements(golden):
    """A measurement outside Y is projected onto Y before it reaches the observer (SSM/controllers.py:96-97)."""
    import sofacontrol_amd.SSM.controllers as sctl
    from sofacontrol_amd.utils import Polyhedron, vq2qv
    model = ossm.synthetic(6, 4, 3, 3, seed=150)
    s = product_ssm(model)
    lim = 0.02
    zr = vq2qv(model['z_ref'])                         # the measurement arrives in [v; q] order
    Y = Polyhedron(np.kron(np.eye(6), np.array([[1.], [-1.]])), np.ravel(np.column_stack((zr + lim, -(zr - lim)))),
                   with_reproject=True)
    c = sctl.scp(s, None, 0.01, N_replan=2, delay=0.0, client=FakeGuSTOClient(), Y=Y)
    c.set_sim_timestep(0.01)
    y = zr + np.array([0.05, -0.01, 0.0, -0.07, 0.015, 0.3])
    quiet(c.evaluate, 0.0, y, None, np.zeros(4))
    np.testing.assert_allclose(vq2qv(c.observer.z), np.clip(y, zr - lim, zr + lim), rtol=0, atol=1e-9)


def test_ssm_closed_loop_with_solver_node():
    """End to end without stand-ins: SSM plant -> measurement -> SSMObserver (W_map on the device) -> `scp` controller ->
    in-process GuSTOClient -> GuSTOSolverNode over SSMGuSTO (device QP with the per-stage observer linearisation) ->
    input back into the plant.  The tracked outputs must approach the target and the inputs respect their box."""
    import sofacontrol_amd.SSM.controllers as sctl
    from sofacontrol_amd.scp.models.ssm import SSMGuSTO
    from sofacontrol_amd.scp.standalone import GuSTOSolverNode
    from sofacontrol_amd.utils import HyperRectangle, vq2qv
    n, m, N, dt = 4, 2, 8, 0.02
    model = ossm.synthetic(n, m, 3, 2, seed=81)
    # a consistent pair of maps (the synthetic coefficients are independent random polynomials): z = x + z_ref and back
    model['W'][:] = 0.0; model['W'][:, :n] = np.eye(n)
    model['V'][:] = 0.0; model['V'][:, :n] = np.eye(n)
    s = product_ssm(model, discr='fe')
    gm = SSMGuSTO(s)
    x = np.zeros(n)
    zf = lambda xx: ossm.observe(model, xx) + model['z_ref']          # oracle C_map is without z_ref
    z_goal = zf(x) + np.array([0.08, -0.04, 0.0, 0.0])
    Qz = np.diag([10., 10., 0.1, 0.1]); R = 1e-2 * np.eye(m)
    U = HyperRectangle([2.0] * m, [-2.0] * m)
    # the solver's cost compares C_map(x) (no z_ref) with its target (locp.py:231-245 with the SSM observer map)
    node = quiet(GuSTOSolverNode, gm, N, dt, Qz, R, x, z=z_goal - model['z_ref'], U=U, verbose=0, max_gusto_iters=4,
                 convg_thresh=1e-4)
    c = sctl.scp(s, None, dt, N_replan=2, delay=0.0, solver_node=node, wait=False)
    c.set_sim_timestep(dt)
    err0 = np.linalg.norm((zf(x) - z_goal)[:2])
    u = np.zeros(m)
    for k in range(24):
        y = vq2qv(zf(x))                                      # the simulator hands over [v; q]
        u = quiet(c.evaluate, k * dt, y, None, u)
        assert np.all(np.abs(u) <= 2.0 + 1e-9)
        np.testing.assert_allclose(c.observer.x, x, rtol=0, atol=1e-9)      # W_map inverts the observation on the manifold
        A, B, d = ossm.jacobians(model, x, u, dt, 'fe')
        x = A @ x + B @ u + d
    err1 = np.linalg.norm((zf(x) - z_goal)[:2])
    assert err1 < 0.5 * err0, (err0, err1)
    info = c.save_controller_info()
    assert len(info['solve_times']) == 12 and info['t_opt'][-1] == pytest.approx(24 * dt)
