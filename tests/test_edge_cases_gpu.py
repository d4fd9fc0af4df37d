"""GPU edge cases along the hot path: empty / single-row / ragged (non tile-multiple) shapes and the smallest
problems each kernel accepts, each against the oracle."""
import io
import contextlib

import os

import numpy as np
import pytest

from oracle import pod as opod, tpwl as otpwl, locp as olocp, lqr as olqr
from helpers import golden_problem, product_tpwl, small_rom, Poly

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-11):
    np.testing.assert_allclose(a, b, rtol=0, atol=rtol * max(1.0, float(np.abs(b).max()) if np.size(b) else 1.0))


def quiet(fn, *a):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a)


@pytest.mark.parametrize('n_nodes,r', [(1, 1), (7, 3), (43, 17), (100, 33)])
def test_projection_ragged_shapes(n_nodes, r):
    from sofacontrol_amd.mor.pod import POD
    U, q_ref, v_ref = small_rom(n_nodes, r, 5)
    rom = POD(dict(U=U, q_ref=q_ref, v_ref=v_ref))
    rng = np.random.default_rng(1)
    for B in (0, 1, 2, 129):
        X = q_ref + rng.standard_normal((B, 3 * n_nodes))
        got = rom.compute_RO_state(qf=X)
        assert got.shape == (B, r)
        if B:
            close(got, opod.project(U, q_ref, X))
            close(rom.compute_FO_state(q=got), got @ U.T + q_ref)
    x = np.concatenate((v_ref, q_ref)) + rng.standard_normal(6 * n_nodes)
    V = np.kron(np.eye(2), U)
    close(rom.compute_RO_state(xf=x), V.T @ (x - np.concatenate((v_ref, q_ref))))
    with pytest.raises(RuntimeError):
        rom.compute_RO_state()


@pytest.mark.parametrize('n_s,n_f', [(1, 5), (3, 200), (130, 77), (257, 1000), (4200, 1100)])   # last: K-split tail tiles
def test_gramian_ragged(n_s, n_f):
    from sofacontrol_amd.mor.pod import gramian
    S = np.random.default_rng(2).standard_normal((n_s, n_f))
    G = gramian(S)
    close(G, S @ S.T, 1e-12)
    np.testing.assert_array_equal(G, G.T)


def test_tpwl_single_point_and_empty_rollout():
    model, U, q_ref, v_ref, Hf = golden_problem(3, 2, 1, 10, 8)
    tp = product_tpwl(model, U, q_ref, v_ref, Hf)
    quiet(tp.pre_discretize, 0.05)
    x = np.random.default_rng(0).standard_normal(6)
    assert tp.calc_nearest_point(x) == 0
    A, B, d = tp.get_jacobians(x, dt=0.05)
    np.testing.assert_array_equal(A, tp.A_d[0])
    X, Z = tp.rollout(x, np.zeros((0, 2)), 0.05)
    assert X.shape == (1, 6) and Z.shape == (1, 6)
    np.testing.assert_array_equal(X[0], x)
    close(Z[0], tp.H @ x + tp.z_ref)
    assert tp.calc_nearest_point(np.zeros((0, 6))).shape == (0,)


def test_locp_smallest_problem():
    """N = 1, n_u = 1, n_z = 1, one U row, one X row, trust region on."""
    from sofacontrol_amd.scp.locp import LOCP
    rng = np.random.default_rng(4)
    n = 2
    A = np.array([[0.9, 0.1], [0.0, 0.8]]); B = np.array([[0.0], [0.5]]); d = np.array([0.01, 0.0])
    H = np.array([[1.0, 0.0]]); Qz = np.array([[10.0]]); R = np.array([[1e-2]])
    x0 = np.array([0.1, -0.2]); xk = np.stack([x0, A @ x0 + d]); z = np.array([[0.0], [0.3]])
    U = (np.array([[1.0]]), np.array([0.4])); X = (np.array([[0.0, 1.0]]), np.array([0.05]))
    qp = olocp.build_qp(1, H, Qz, R, [A], [B], [d], x0, xk, 0.5, 2.0, z=z, U=U, X=X)
    w, _, info = olocp.solve_exact(qp, tol=1e-12)
    xe, ue, se = olocp.split(qp, w)
    locp = LOCP(1, H, Qz, R, U=Poly(*U), X=Poly(*X))
    locp.update([A], [B], [d], x0, xk, 0.5, 2.0, z=z)
    J, ok, _ = locp.solve()
    assert ok
    x, u, s = locp.get_solution()
    close(x, xe, 1e-7); close(u, ue, 1e-7)
    assert abs(J - olocp.objective(qp, w)) <= 1e-8 * max(1.0, abs(J))


def test_gusto_zero_iterations_limit(golden):
    """max_gusto_iters = 0 after the constructor solve: exactly one QP per call (gusto.py:163-172)."""
    from sofacontrol_amd.scp.models.tpwl import TPWLGuSTO
    from sofacontrol_amd.scp.gusto import GuSTO
    g = golden('g6_gusto')
    model, U, q_ref, v_ref, Hf = golden_problem(4, 3, 7, 20, 30, q_scale=0.05)
    tp = product_tpwl(model, U, q_ref, v_ref, Hf)
    gm = TPWLGuSTO(tp)
    quiet(gm.pre_discretize, 0.05)
    N, dt = 12, 0.05
    x0 = np.zeros(8); u_init = np.zeros((N, 3))
    x_init, _ = gm.rollout(x0, u_init, dt)
    from scipy.interpolate import interp1d
    z = interp1d(g['t'], g['zt'], axis=0)(dt * np.arange(N + 1))
    gu = GuSTO(gm, N, dt, g['Qz'], g['R'], x0, u_init, x_init, z=z, x_char=g['x_char'], f_char=g['f_char'],
               convg_thresh=1e-3, max_gusto_iters=0, U=Poly(g['U_A'], g['U_b']))
    gu.solve(x0, u_init, x_init, z=z)
    assert int(gu.iters[0]) == 1


def test_ilqr_horizon_one():
    from sofacontrol_amd.lqr.ilqr import iLQR
    from sofacontrol_amd.utils import QuadraticCost
    model, U, q_ref, v_ref, Hf = golden_problem(3, 2, 4, 10, 9, q_scale=0.1)
    tp = product_tpwl(model, U, q_ref, v_ref, Hf)
    quiet(tp.pre_discretize, 0.05)
    Ad, Bd, dd = np.stack(tp.A_d), np.stack(tp.B_d), np.stack(tp.d_d)
    Qz = np.diag([0., 0., 0., 50., 50., 5.]); R = 1e-2 * np.eye(2)
    zt = np.asarray(tp.z_ref) + np.array([[0, 0, 0, 0.01, 0.0, 0], [0, 0, 0, 0.02, -0.01, 0]])
    x0 = 1e-3 * np.ones(6)
    il = iLQR(0.05, tp, QuadraticCost(Q=Qz, R=R, Qf=Qz), 1)
    il.set_target(zt)
    x, u, K = il.ilqr_computation(x0)
    o = olqr.ILQR(model, Ad, Bd, dd, np.asarray(tp.H), np.asarray(tp.z_ref), Qz, R, Qz, 1)
    xo, uo, Ko = o.solve(x0, zt)
    assert int(il.iters[0]) == len(o.trace) - 1
    close(x, xo, 1e-8); close(u, uo, 1e-7)


def test_errors_are_loud():
    """Bad arguments come back as HipError (RuntimeError) with the C ABI's message, never a silent result."""
    from sofacontrol_amd import _lib
    from sofacontrol_amd.mor.pod import POD
    U, q_ref, v_ref = small_rom(5, 2, 1)
    rom = POD(dict(U=U, q_ref=q_ref, v_ref=v_ref))
    with pytest.raises(Exception):
        rom.compute_RO_state(qf=np.zeros(7))          # wrong length
    model, U, q_ref, v_ref, Hf = golden_problem(3, 2, 4, 10, 9)
    tp = product_tpwl(model, U, q_ref, v_ref, Hf)
    from sofacontrol_amd.lqr.ilqr import iLQR
    from sofacontrol_amd.utils import QuadraticCost, Polyhedron
    from sofacontrol_amd.scp.locp import LOCP
    with pytest.raises(ValueError, match='dU'):
        LOCP(3, np.eye(2), np.eye(2), np.eye(1), dU=Polyhedron(np.array([[1.0], [-1.0]]), -np.ones(2)))   # empty rate polyhedron
    locp = LOCP(3, np.eye(2), np.eye(2), np.eye(1))
    locp.update([np.eye(2)] * 3, [np.ones((2, 1))] * 3, [np.zeros(2)] * 3, np.zeros(2), None, 1.0, 1.0)
    with pytest.raises(RuntimeError, match='xk'):
        locp.solve()                                   # trust region active without its centre
    # (the iLQR configuration switches no longer raise: they are kernel parameters since round 4, tests/test_lqr_gpu.py)
    il = iLQR(0.05, tp, QuadraticCost(Q=np.eye(6), R=np.eye(2), Qf=np.eye(6)), 5)
    with pytest.raises(Exception):
        il.ilqr_computation(np.zeros(6))               # no target set


@pytest.mark.parametrize('n', [1, 2, 5, 40, 127, 128, 129, 150, 257, 400])
def test_device_eigh(n):
    """srom_eigh_dev: one-workgroup Jacobi in LDS for n <= 128, the same Jacobi over HBM (two launches per round-robin
    step) up to 2048 -- odd sizes pad with a dummy row that never mixes."""
    from sofacontrol_amd.mor.pod import _device_eigh
    rng = np.random.default_rng(n)
    S = rng.standard_normal((n, n + 3)) * np.logspace(0, -3, n + 3)
    G = S @ S.T
    w, W = _device_eigh(G)
    we = np.linalg.eigvalsh(G)
    np.testing.assert_allclose(w, we, rtol=0, atol=1e-12 * max(1.0, np.abs(we).max()))
    np.testing.assert_allclose(W.T @ W, np.eye(n), atol=1e-12)
    np.testing.assert_allclose(G @ W, W * w, atol=1e-11 * max(1.0, np.abs(we).max()))


def test_device_eigh_rocsolver_path():
    """n > 2048 goes to rocSOLVER dsyevd (dlopen at first use); SRH_EIGH_ROCSOLVER forces it for any n.  (The first
    rocSOLVER / rocBLAS load of a process pages the libraries in: seconds on a warm box, up to minutes on a cold one.)"""
    from sofacontrol_amd.mor.pod import _device_eigh
    os.environ['SRH_EIGH_ROCSOLVER'] = '1'
    try:
        n = 150
        rng = np.random.default_rng(n)
        S = rng.standard_normal((n, n + 3)) * np.logspace(0, -3, n + 3)
        G = S @ S.T
        w, W = _device_eigh(G)
    finally:
        del os.environ['SRH_EIGH_ROCSOLVER']
    we = np.linalg.eigvalsh(G)
    np.testing.assert_allclose(w, we, rtol=0, atol=1e-12 * max(1.0, np.abs(we).max()))
    np.testing.assert_allclose(G @ W, W * w, atol=1e-11 * max(1.0, np.abs(we).max()))


def test_device_eigh_dispatch_boundaries():
    """The size-based dispatch on both sides of its last boundary: 4096 snapshots run the block Jacobi (pairs of 64), 4100 the library
    (rocSOLVER -- loaded by the test above in this process -- or, where it does not load, the block Jacobi); leading pairs and the spectrum
    against numpy."""
    from sofacontrol_amd.mor.pod import _device_eigh
    for n in (4096, 4100):
        rng = np.random.default_rng(n)
        S = rng.standard_normal((n, 96)) * np.logspace(0, -2, 96)
        G = S @ S.T + 1e-6 * np.eye(n)
        w, W = _device_eigh(G)
        we = np.linalg.eigvalsh(G)
        np.testing.assert_allclose(w, we, rtol=0, atol=2e-11 * np.abs(we).max())
        np.testing.assert_allclose(G @ W[:, -8:], W[:, -8:] * w[-8:], atol=1e-10 * np.abs(we).max())
        np.testing.assert_allclose(W[:, -32:].T @ W[:, -32:], np.eye(32), atol=5e-11)


def test_device_eigh_above_jacobi_limit():
    """n = 2100 > 2048, the scalar Jacobi's limit: the size-based dispatch itself (no environment override) picks the block Jacobi up to
    4096 snapshots, rocSOLVER above (the block Jacobi if the library does not load)."""
    from sofacontrol_amd.mor.pod import _device_eigh
    n = 2100
    rng = np.random.default_rng(n)
    S = rng.standard_normal((n, 64)) * np.logspace(0, -2, 64)
    G = S @ S.T + 1e-6 * np.eye(n)
    w, W = _device_eigh(G)
    we = np.linalg.eigvalsh(G)
    np.testing.assert_allclose(w, we, rtol=0, atol=1e-11 * np.abs(we).max())
    np.testing.assert_allclose(G @ W[:, -8:], W[:, -8:] * w[-8:], atol=1e-10 * np.abs(we).max())


@pytest.mark.parametrize('n', [129, 300, 1000, 2100])
def test_device_eigh_block_jacobi(n):
    """SRH_EIGH_BLOCK=1 (and any box where rocSOLVER does not load): the library-free two-sided block Jacobi of csrc/eigh.hip --
    64-wide blocks, 128 x 128 pair problems in LDS, the step applied as chained MFMA products.  Full spectrum and eigenvectors
    against numpy (mor/pod.py:181-200 returns the whole spectrum); sizes that are not multiples of 128 pad with rows that never mix."""
    from sofacontrol_amd.mor.pod import _device_eigh
    rng = np.random.default_rng(n)
    k = min(n + 3, 700)                                   # (n = 1000, 2100: rank-deficient -- a 300+-fold zero eigenvalue)
    S = rng.standard_normal((n, k)) * np.logspace(0, -3, k)
    G = S @ S.T
    os.environ['SRH_EIGH_BLOCK'] = '1'
    try:
        w, W = _device_eigh(G)
    finally:
        del os.environ['SRH_EIGH_BLOCK']
    we = np.linalg.eigvalsh(G)
    scale = np.abs(we).max()
    assert np.all(np.diff(w) >= 0)
    np.testing.assert_allclose(w, we, rtol=0, atol=1e-11 * scale)
    np.testing.assert_allclose(W.T @ W, np.eye(n), atol=2e-11)
    np.testing.assert_allclose(G @ W, W * w, atol=1e-10 * scale)


def test_new_entry_points_reject_bad_arguments():
    """Round-2 entry points: argument checks raise (never a silent fallback), empty batches are no-ops."""
    import ctypes as C
    from sofacontrol_amd import _lib
    from sofacontrol_amd.utils import Polyhedron
    from sofacontrol_amd.lqr.lqr import dare
    L = _lib.lib()
    A, b = np.vstack((np.eye(2), -np.eye(2))), np.array([1.0, 1.0, -2.0, 1.0])      # x_0 <= 1 and x_0 >= 2: empty set
    with pytest.raises(Exception, match='spoly_project'):
        Polyhedron(A, b, with_reproject=True).project_to_polyhedron(np.array([5.0, 0.0]))
    P = Polyhedron(np.vstack((np.eye(2), -np.eye(2))), np.ones(4), with_reproject=True)
    assert P.project_to_polyhedron(np.zeros((0, 2))).shape == (0, 2)
    with pytest.raises(RuntimeError):
        P.project_to_polyhedron(np.zeros(3))
    big = Polyhedron(np.vstack((np.eye(17), -np.eye(17))), np.ones(34), with_reproject=True)
    with pytest.raises(RuntimeError):
        big.project_to_polyhedron(2 * np.ones(17))                                  # n <= 16
    with pytest.raises(RuntimeError):
        dare(np.eye(3), np.ones((3, 17)), np.eye(3), np.eye(17))                       # n_u <= 16
    with pytest.raises(Exception):
        dare(np.eye(2), np.ones((2, 1)), np.eye(2), -np.eye(1))                        # R not positive definite
    x = np.zeros(4)
    rc = L.sekf_step_projected(None, None, _lib.dptr(x), None, None, _lib.dptr(x), None)
    assert rc != 0 and b'sekf_step_projected' in L.srh_last_error()
