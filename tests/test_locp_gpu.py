"""GPU parity: the LOCP QP kernel through the C ABI against the exact oracle solution.
Tolerance (north star): <= 1e-4 relative on trajectories, 1e-7 relative on the optimal cost.

Both sides stop at the same complementarity gap (1e-12): with R = 1e-5 against Qz = 100 and weakly
active bounds the minimiser is only determined to ~1e-3 in the flat input directions by ANY solver at
that gap ("QP conditioning"), so the comparison is made at equal tolerance, where the
independent sparse solver and the kernel land on the same central-path point."""
import numpy as np
import pytest

from oracle import locp as olocp
from qp_cases import CASES, make_case
from helpers import Poly

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def oracle_solution(case):
    qp = olocp.build_qp(case['N'], case['H'], case['Qz'], case['R'], case['Ad'], case['Bd'], case['dd'], case['x0'],
                        case['xk'], case['delta'], case['omega'], z=case['z'], Qzf=case.get('Qzf'), zf=case.get('zf'),
                        U=case['U'], X=case['X'], x_scale=case['x_scale'])
    w, _, info = olocp.solve_exact(qp, tol=1e-12)
    assert info.get('status', 'optimal') == 'optimal'
    return olocp.split(qp, w), olocp.objective(qp, w)


def same_path_reference(case):
    """The oracle run the way the kernel runs: the QP without its trust-region rows first (its minimiser is the minimiser of
    the full QP whenever it lies inside the trust region -- locp_dev.h, oracle/riccati_ipm.py), stopped at the same gap.  Two
    exact solvers agree on these flat QPs (R = 1e-5, weakly active bounds) only to ~1e-4 in the trajectory when they follow
    DIFFERENT central paths; along the same path the kernel is held to 1e-9."""
    from oracle import riccati_ipm as ripm
    xr, ur, sr, Jr, info = ripm.solve(ripm.Problem(**dict(case, tr_active=False)))
    assert info['status'] == 'optimal'
    assert np.abs(case['x_scale'] * (xr[1:] - case['xk'][1:])).max() <= case['delta']      # inside: it IS the full QP's minimiser
    return xr, ur


def product_locp(case):
    from sofacontrol_amd.scp.locp import LOCP
    U = Poly(*case['U']) if case['U'] is not None else None
    X = Poly(*case['X']) if case['X'] is not None else None
    return LOCP(case['N'], case['H'], case['Qz'], case['R'], Qzf=case.get('Qzf'), U=U, X=X,
                x_char=1. / case['x_scale'])


@pytest.mark.parametrize('name', list(CASES))
def test_locp_matches_exact_solution(name):
    case, _ = make_case(**CASES[name])
    (xe, ue, se), Je = oracle_solution(case)
    locp = product_locp(case)
    locp.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], case['delta'],
                case['omega'], z=case['z'], zf=case.get('zf'))
    J, ok, stats = locp.solve()
    assert ok
    x, u, s = locp.get_solution()
    assert rel(x, xe) <= 1e-4 and rel(u, ue) <= 1e-4
    assert abs(J - Je) <= 1e-7 * max(1.0, abs(Je))
    np.testing.assert_allclose(s, se, rtol=0, atol=1e-4 * max(1.0, se.max()))
    # dynamics hold exactly along the returned trajectory (equality constraints, locp.py:287, 340)
    np.testing.assert_array_equal(x[0], case['x0'])
    for k in range(case['N']):
        np.testing.assert_allclose(x[k + 1], case['Ad'][k] @ x[k] + case['Bd'][k] @ u[k] + case['dd'][k], rtol=0, atol=1e-12)
    # update(full=False) only changes delta / omega (locp.py:139-141)
    locp.update(None, None, None, None, None, case['delta'], case['omega'], full=False)
    J2, ok2, _ = locp.solve()
    assert ok2 and J2 == J


def test_locp_diamond_shape():
    """C2 shape (n_x = 60, n_u = 4, N = 50): KKT feasibility + optimality vs the exact oracle."""
    case, _ = make_case(r=30, m=4, P=64, N=50, seed=7, q_scale=0.02, use_X=True, u_max=1500.0, amp=0.1)
    (xe, ue, se), Je = oracle_solution(case)
    locp = product_locp(case)
    locp.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], case['delta'],
                case['omega'], z=case['z'])
    J, ok, stats = locp.solve()
    assert ok
    x, u, s = locp.get_solution()
    assert abs(J - Je) <= 1e-7 * max(1.0, abs(Je))
    assert rel(x, xe) <= 1e-4 and rel(u, ue) <= 1e-4
    UA, Ub = case['U']
    assert (UA @ u.T - Ub[:, None]).max() <= 1e-7


def test_infeasible_qp_reports_failure():
    """locp.py:187-190: a QP that cannot be solved returns (inf, False, None)."""
    case, _ = make_case(**CASES['box_X'])
    case['X'] = (case['X'][0], np.array([-1.0, -1.0, -1.0, -1.0]))    # empty set
    locp = product_locp(case)
    locp.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], case['delta'],
                case['omega'], z=case['z'])
    J, ok, stats = locp.solve()
    assert not ok and J == np.inf and stats is None


@pytest.mark.parametrize('use_X', [False, True])
def test_locp_trunk_shape(use_X):
    """C5 shape (Trunk: n_x = 60, n_u = 8, N = 50, U = [0, 800]^8; examples/trunk/trunk.py:309-316 has X = None,
    the X-box variant exercises the LDS budget): the n_x + n_u = 68 panels (80-wide tiles)."""
    case, _ = make_case(r=30, m=8, P=32, N=50, seed=11, q_scale=0.02, use_X=use_X, u_max=800.0, amp=0.1, x_box=4.0)
    (xe, ue, se), Je = oracle_solution(case)
    locp = product_locp(case)
    locp.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], case['delta'],
                case['omega'], z=case['z'])
    J, ok, stats = locp.solve()
    assert ok
    x, u, s = locp.get_solution()
    assert abs(J - Je) <= 1e-7 * max(1.0, abs(Je))
    # along the oracle's own central path (the trust-region-free relaxation, as the kernel): 1e-9
    xr, ur = same_path_reference(case)
    assert rel(x, xr) <= 1e-9 and rel(u, ur) <= 1e-9
    # against the exact sparse solver (another central path): the north star's 1e-4 on the trajectory; the inputs of this
    # flat QP (8 inputs, R = 1e-5, weakly active bounds) differ by up to 1.03e-4 BETWEEN the two oracles themselves
    assert rel(x, xe) <= 1e-4 and rel(u, ue) <= 1.5e-4, (rel(x, xe), rel(u, ue))
    from oracle import riccati_ipm as ripm
    xp, up, sp, Jp, info = ripm.solve(ripm.Problem(**case))
    assert info['status'] == 'optimal'
    assert rel(x, xp) <= 1e-4 and rel(u, up) <= 1.5e-4 and abs(J - Jp) <= 1e-7 * max(1.0, abs(Jp))


@pytest.mark.parametrize('terminal', [False, True])
def test_linear_mpc_no_trust_region(terminal):
    """LOCP(is_tr_active=False) (locp.py:57; the linear-MPC baselines of baselines/ros.py) and the MPCSolver
    mirror: constant (A, B, d), U box + X box, optional terminal cost, vs the exact oracle solution."""
    from sofacontrol_amd.baselines.mpc import MPCSolver
    from sofacontrol_amd.utils import QuadraticCost
    from sofacontrol_amd.tpwl.tpwl_utils import Target
    case, extra = make_case(seed=41, terminal=terminal, N=15)
    A, B, d = extra['Ad_tab'][2], extra['Bd_tab'][2], extra['dd_tab'][2]
    N = case['N']
    qp = olocp.build_qp(N, case['H'], case['Qz'], case['R'], [A] * N, [B] * N, [d] * N, case['x0'], None, 0, 0,
                        z=case['z'], Qzf=case.get('Qzf'), zf=case.get('zf'), U=case['U'], X=case['X'], tr_active=False)
    w, _, info = olocp.solve_exact(qp, tol=1e-12)
    xe, ue, _ = olocp.split(qp, w)
    Je = olocp.objective(qp, w)

    class M:
        H = case['H']; A_d = A; B_d = B; d_d = d
    tgt = Target(); tgt.t = extra['dt'] * np.arange(N + 1); tgt.z = case['z']; tgt.u = None
    mpc = MPCSolver(M, N, extra['dt'], QuadraticCost(Q=case['Qz'], R=case['R'], Qf=case.get('Qzf')), case['x0'], tgt,
                    U=Poly(*case['U']), X=Poly(*case['X']))
    x, u, z, t = mpc.get_solution()
    assert rel(x, xe) <= 1e-4 and rel(u, ue) <= 1e-4
    assert abs(mpc.Jstar - Je) <= 1e-7 * max(1.0, abs(Je))
    assert mpc.locp.get_solution()[2] is None            # no slack variables without the trust region
    np.testing.assert_allclose(z, x @ case['H'].T, atol=1e-14)


@pytest.mark.parametrize('m,use_X', [(4, True), (4, False), (8, False)])
def test_locp_r36_split_panel(m, use_X):
    """n_x = 72 (the POD basis the reference ships for the Diamond at tol 5e-5 has r = 36): P, [A|B] and W no longer
    fit LDS together; the kernel produces W 48 rows at a time and accumulates the Gram products in MFMA registers."""
    case, _ = make_case(r=36, m=m, P=16, N=30, seed=17, q_scale=0.02, use_X=use_X, u_max=800.0, amp=0.1, x_box=4.0)
    (xe, ue, se), Je = oracle_solution(case)
    locp = product_locp(case)
    locp.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], case['delta'],
                case['omega'], z=case['z'])
    J, ok, stats = locp.solve()
    assert ok
    x, u, s = locp.get_solution()
    assert abs(J - Je) <= 1e-7 * max(1.0, abs(Je))
    xr, ur = same_path_reference(case)
    assert rel(x, xr) <= 1e-9 and rel(u, ur) <= 1e-9
    assert rel(x, xe) <= 1e-4 and rel(u, ue) <= 1e-4
    for k in range(case['N']):
        np.testing.assert_allclose(x[k + 1], case['Ad'][k] @ x[k] + case['Bd'][k] @ u[k] + case['dd'][k], rtol=0, atol=1e-12)


def test_locp_terminal_set_and_input_target():
    """Xf rows on x_N (locp.py:336-337) and an input target u_des (locp.py:226) together with U, X, Qzf."""
    from sofacontrol_amd.scp.locp import LOCP
    case, _ = make_case(seed=47, terminal=True, N=14, x_box=2.0)
    XA, Xb = case['X']
    Xf = (XA[:2], 0.5 * Xb[:2])                       # tighter box on the terminal state
    rng = np.random.default_rng(8)
    u_des = rng.uniform(0, 30, (case['N'], case['Bd'][0].shape[1]))
    qp = olocp.build_qp(case['N'], case['H'], case['Qz'], case['R'], case['Ad'], case['Bd'], case['dd'], case['x0'],
                        case['xk'], case['delta'], case['omega'], z=case['z'], u_des=u_des, Qzf=case['Qzf'],
                        zf=case['zf'], U=case['U'], X=case['X'], Xf=Xf, x_scale=case['x_scale'])
    w, _, info = olocp.solve_exact(qp, tol=1e-12)
    assert info.get('status', 'optimal') == 'optimal'
    xe, ue, se = olocp.split(qp, w)
    Je = olocp.objective(qp, w)
    locp = LOCP(case['N'], case['H'], case['Qz'], case['R'], Qzf=case['Qzf'], U=Poly(*case['U']), X=Poly(*case['X']),
                Xf=Poly(*Xf), x_char=1. / case['x_scale'])
    locp.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], case['delta'],
                case['omega'], z=case['z'], zf=case['zf'], u=u_des)
    J, ok, _ = locp.solve()
    assert ok
    x, u, s = locp.get_solution()
    assert rel(x, xe) <= 1e-4 and rel(u, ue) <= 1e-4
    assert abs(J - Je) <= 1e-7 * max(1.0, abs(Je))
    assert np.all(Xf[0] @ x[-1] <= Xf[1] + 1e-9)


@pytest.mark.parametrize('shape', ['small', 'diamond'])
def test_locp_input_rate_constraints(shape):
    """dU.A (u_{k+1} - u_k) <= dU.b (locp.py:305-308) through the state augmentation [x; u_prev; du] (the Diamond size
    lands in the split-panel kernel: n_x + 2 n_u = 68)."""
    from sofacontrol_amd.scp.locp import LOCP
    kw = dict(seed=52, N=12) if shape == 'small' else dict(r=30, m=4, P=16, N=20, seed=53, q_scale=0.02, u_max=1500.0,
                                                            amp=0.1, x_box=4.0)
    case, _ = make_case(**kw)
    m = case['Bd'][0].shape[1]
    dA = np.kron(np.eye(m), np.array([[1.], [-1.]]))
    db = np.full(2 * m, 8.0 if shape == 'small' else 5.0)
    qp = olocp.build_qp(case['N'], case['H'], case['Qz'], case['R'], case['Ad'], case['Bd'], case['dd'], case['x0'],
                        case['xk'], case['delta'], case['omega'], z=case['z'], U=case['U'], X=case['X'], dU=(dA, db),
                        x_scale=case['x_scale'])
    w, _, info = olocp.solve_exact(qp, tol=1e-12)
    assert info.get('status', 'optimal') == 'optimal'
    xe, ue, se = olocp.split(qp, w)
    Je = olocp.objective(qp, w)
    assert np.abs(np.diff(ue, axis=0)).max() > 0.99 * db[0]          # the rate limit is active
    locp = LOCP(case['N'], case['H'], case['Qz'], case['R'], U=Poly(*case['U']), X=Poly(*case['X']), dU=Poly(dA, db),
                x_char=1. / case['x_scale'])
    locp.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], case['delta'],
                case['omega'], z=case['z'])
    J, ok, _ = locp.solve()
    assert ok
    x, u, s = locp.get_solution()
    assert x.shape == xe.shape
    assert rel(x, xe) <= 1e-4 and rel(u, ue) <= 1e-4
    assert abs(J - Je) <= 1e-7 * max(1.0, abs(Je))
    assert np.abs(np.diff(u, axis=0)).max() <= db[0] * (1 + 1e-6)


def test_locp_input_rate_constraints_with_negative_bounds():
    """dU.b with negative entries (every input has to rise by at least 0.3 per step): the zero increment is infeasible, the
    augmented dynamics start from an interior point of the rate polyhedron instead (locp.py:305-308)."""
    from sofacontrol_amd.scp.locp import LOCP
    case, _ = make_case(seed=57, N=10)
    m = case['Bd'][0].shape[1]
    dA = np.kron(np.eye(m), np.array([[1.], [-1.]]))
    db = np.tile(np.array([6.0, -0.3]), m)                         # 0.3 <= u_{k+1} - u_k <= 6
    U = (case['U'][0], case['U'][1] + 100.0)                       # room for ten rising steps
    qp = olocp.build_qp(case['N'], case['H'], case['Qz'], case['R'], case['Ad'], case['Bd'], case['dd'], case['x0'],
                        case['xk'], case['delta'], case['omega'], z=case['z'], U=U, dU=(dA, db), x_scale=case['x_scale'])
    w, _, info = olocp.solve_exact(qp, tol=1e-12)
    assert info.get('status', 'optimal') == 'optimal'
    xe, ue, se = olocp.split(qp, w)
    assert np.diff(ue, axis=0).min() < 0.3 * (1 + 1e-6)            # the lower rate bound is active somewhere
    locp = LOCP(case['N'], case['H'], case['Qz'], case['R'], U=Poly(*U), dU=Poly(dA, db), x_char=1. / case['x_scale'])
    locp.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], case['delta'], case['omega'], z=case['z'])
    J, ok, _ = locp.solve()
    assert ok
    x, u, s = locp.get_solution()
    assert rel(x, xe) <= 1e-4 and rel(u, ue) <= 1e-4 and abs(J - olocp.objective(qp, w)) <= 1e-7 * max(1.0, abs(olocp.objective(qp, w)))
    assert np.diff(u, axis=0).min() >= 0.3 * (1 - 1e-6)
    with pytest.raises(ValueError, match='no interior'):
        LOCP(3, np.eye(2), np.eye(2), np.eye(1), dU=Poly(np.array([[1.0], [-1.0]]), np.array([-1.0, -1.0])))   # e >= 1 and e <= -1


def test_locp_input_rate_constraints_with_nonlinear_observer():
    """dU together with the per-stage observer linearisation (locp.py:231-245, 305-308, 312-329): both augmentations at once,
    xa = [x; zeta; u_prev; du]."""
    from sofacontrol_amd.scp.locp import LOCP
    case, _ = make_case(seed=58, N=9)
    N, n = case['N'], case['Ad'][0].shape[0]
    m, nz = case['Bd'][0].shape[1], case['H'].shape[0]
    rng = np.random.default_rng(58)
    Hd = case['H'][None] + 0.05 * rng.standard_normal((N + 1, nz, n))
    cd = 0.02 * rng.standard_normal((N + 1, nz))
    dA = np.kron(np.eye(m), np.array([[1.], [-1.]]))
    db = np.full(2 * m, 5.0)
    Xz = (np.vstack((np.eye(nz), -np.eye(nz))), np.full(2 * nz, 50.0))          # rows on the outputs (locp.py:312-329)
    qp = olocp.build_qp(N, np.zeros_like(case['H']), case['Qz'], case['R'], case['Ad'], case['Bd'], case['dd'], case['x0'], case['xk'],
                        case['delta'], case['omega'], z=case['z'], U=case['U'], X=Xz, dU=(dA, db), x_scale=case['x_scale'], Hd=Hd, cd=cd)
    w, _, info = olocp.solve_exact(qp, tol=1e-12)
    assert info.get('status', 'optimal') == 'optimal'
    xe, ue, se = olocp.split(qp, w)
    assert np.abs(np.diff(ue, axis=0)).max() > 0.99 * db[0]
    locp = LOCP(N, np.zeros_like(case['H']), case['Qz'], case['R'], U=Poly(*case['U']), X=Poly(*Xz), dU=Poly(dA, db),
                x_char=1. / case['x_scale'], nonlinear_observer=True)
    locp.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], case['delta'], case['omega'],
                z=case['z'], Hd=list(Hd), cd=list(cd))
    J, ok, _ = locp.solve()
    assert ok
    x, u, s = locp.get_solution()
    assert x.shape == xe.shape and rel(x, xe) <= 1e-4 and rel(u, ue) <= 1e-4
    assert abs(J - olocp.objective(qp, w)) <= 1e-7 * max(1.0, abs(olocp.objective(qp, w)))


def test_resident_locp_plan_keeps_the_horizon_and_takes_device_pointers():
    """slocp_plan_* (round 3): LOCP.update(full=False) (locp.py:139-141) re-solves on the resident horizon -- nothing but delta,
    omega, x0 goes up; slocp_plan_solve_dev takes every array from HBM and leaves the results there.  Both give what the
    one-shot slocp_solve gives, bit for bit."""
    import ctypes as C
    from sofacontrol_amd import _lib
    from sofacontrol_amd.scp.locp import LOCP, make_problem
    case, _ = make_case(seed=61, N=12)
    N, n, m = case['N'], case['Ad'][0].shape[0], case['Bd'][0].shape[1]

    def fresh(delta, omega):
        lo = LOCP(N, case['H'], case['Qz'], case['R'], U=Poly(*case['U']), X=Poly(*case['X']), x_char=1. / case['x_scale'])
        lo.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], delta, omega, z=case['z'])
        J, ok, st = lo.solve()
        assert ok
        return (J,) + lo.get_solution()

    locp = LOCP(N, case['H'], case['Qz'], case['R'], U=Poly(*case['U']), X=Poly(*case['X']), x_char=1. / case['x_scale'])
    locp.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], case['delta'], case['omega'], z=case['z'])
    J1, ok, _ = locp.solve()
    assert ok
    ref1 = fresh(case['delta'], case['omega'])
    assert J1 == ref1[0] and all(np.array_equal(a, b) for a, b in zip(locp.get_solution(), ref1[1:]))
    used = np.abs(case['x_scale'] * (locp.get_solution()[0] - case['xk'])).max()
    d2, o2 = 0.5 * used, 10.0 * case['omega']                        # a trust region that binds, a larger penalty
    locp.update(None, None, None, None, None, d2, o2, full=False)
    J2, ok, _ = locp.solve()
    assert ok and J2 != J1
    ref2 = fresh(d2, o2)
    assert J2 == ref2[0] and all(np.array_equal(a, b) for a, b in zip(locp.get_solution(), ref2[1:]))
    # device pointers in, device pointers out
    L = _lib.lib()
    prob, keep = make_problem(N, case['H'], case['Qz'], case['R'], None, Poly(*case['U']), Poly(*case['X']), None, None, case['x_scale'], True)
    plan = C.c_void_p()
    _lib.check(L.slocp_plan_create(C.byref(plan), C.byref(prob), C.c_int64(1)), 'create')
    up = lambda a: _lib.DeviceBuffer.from_array(np.ascontiguousarray(a, dtype=np.float64))
    dA, dB, dd, dx0, dxk, dz = up(np.stack(case['Ad'])), up(np.stack(case['Bd'])), up(np.stack(case['dd'])), up(case['x0']), up(case['xk']), up(case['z'])
    ddel, dom = up(np.array([d2])), up(np.array([o2]))
    ox, ou, os_, oJ = _lib.DeviceBuffer((N + 1) * n * 8), _lib.DeviceBuffer(N * m * 8), _lib.DeviceBuffer((N + 1) * 8), _lib.DeviceBuffer(8)
    ost, oit = _lib.DeviceBuffer(4), _lib.DeviceBuffer(4)
    _lib.check(L.slocp_plan_solve_dev(plan, dA.ptr, dB.ptr, dd.ptr, dx0.ptr, dxk.ptr, ddel.ptr, dom.ptr, dz.ptr, None, None,
                                      ox.ptr, ou.ptr, os_.ptr, oJ.ptr, ost.ptr, oit.ptr, None), 'solve_dev')
    _lib.sync()
    assert ost.to_array((1,), dtype=np.int32)[0] == 0 and oJ.to_array((1,))[0] == ref2[0]
    assert np.array_equal(ox.to_array((N + 1, n)), ref2[1]) and np.array_equal(ou.to_array((N, m)), ref2[2])
    L.slocp_plan_destroy(plan)


# ---------------------------------------------------------------------------------------------------------------
# input_nullspace (locp.py:70-71, 258-261): J += || tile(input_nullspace, N) @ u ||_2 -- a norm, not a square
@pytest.mark.parametrize('name', ['vec_smooth', 'vec_kink', 'mat_smooth', 'mat_kink'])
def test_locp_input_nullspace_term(name):
    """LOCP(input_nullspace=...) -- dual maximisation around the device QP (LOCP._solve_nullspace) -- against the optimum that the
    reference's own objective was evaluated at (golden g21, generated through the reference's locp.py) and the oracle's duality
    certificate: the returned multiplier has norm <= 1, the returned inputs minimise the QP with that multiplier's linear cost
    (exact oracle solve), and the gap ||g|| - mu' g is zero -- together sufficient for global optimality."""
    import os
    import qp_cases
    from sofacontrol_amd.scp.locp import LOCP
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g21_locp_nullspace.npz'))
    case, ns = qp_cases.nullspace_case(name)
    kw = dict(case)
    qp = olocp.build_qp(kw.pop('N'), kw.pop('H'), kw.pop('Qz'), kw.pop('R'), kw.pop('Ad'), kw.pop('Bd'), kw.pop('dd'),
                        kw.pop('x0'), kw.pop('xk'), kw.pop('delta'), kw.pop('omega'), **kw)
    xe, ue, se = olocp.split(qp, g[name + '_wopt'])
    Je = float(g[name + '_Jopt'])
    locp = LOCP(case['N'], case['H'], case['Qz'], case['R'], U=Poly(*case['U']), X=Poly(*case['X']), x_char=1. / case['x_scale'],
                input_nullspace=ns)
    locp.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], case['delta'], case['omega'],
                z=case['z'], u=case['u_des'])
    J, ok, stats = locp.solve()
    assert ok
    x, u, s = locp.get_solution()
    st = locp.nullspace_stats
    assert abs(J - Je) <= 1e-7 * max(1.0, abs(Je)), (J, Je, st)
    assert rel(x, xe) <= 1e-4 and rel(u, ue) <= 2e-3                      # (flat input directions: see the module docstring)
    w = qp_cases.g14_pack(case, x, u, s)
    assert J == pytest.approx(olocp.objective(qp, w) + olocp.nullspace_term(qp, ns, w), rel=1e-9)      # J is the reference's objective
    cert = olocp.nullspace_certificate(qp, ns, w, st['mu'])
    assert cert['mu_norm'] <= 1 + 1e-12 and cert['gap'] <= 1e-8 * max(1.0, abs(Je)) and abs(cert['inner_dJ']) <= 1e-6, (cert, st)
    kink = name.endswith('kink')
    assert (st['term'] <= 1e-6) == kink
    assert st['qp_solves'] <= (40 if kink else 3 if name.startswith('vec') else 40), st
    # dynamics hold along the returned trajectory
    for k in range(case['N']):
        np.testing.assert_allclose(x[k + 1], case['Ad'][k] @ x[k] + case['Bd'][k] @ u[k] + case['dd'][k], rtol=0, atol=1e-12)


def test_locp_input_nullspace_argument_checks():
    from sofacontrol_amd.scp.locp import LOCP
    with pytest.raises(ValueError, match='input_nullspace'):
        LOCP(3, np.eye(2), np.eye(2), np.eye(3), input_nullspace=np.ones(2))                 # n_u = 3
    with pytest.raises(ValueError, match='positive definite'):
        LOCP(3, np.eye(2), np.eye(2), np.zeros((3, 3)), input_nullspace=np.ones(3))
