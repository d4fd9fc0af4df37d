"""GPU parity of the condensed (output-space) interior point (csrc/locp_cond.h) -- the path QPs take when their trust
region is inactive -- against its numpy statement oracle.condensed_ipm (same iterates: tight tolerance) and the
independent sparse solver; the general variants (dense input Hessian blocks, three output directions, terminal rows)
and the stage-wise Riccati path that remains behind it (SRH_QP_NO_COND=1, and whenever the minimiser leaves the trust
region)."""
import os

import numpy as np
import pytest

from oracle import locp as olocp, riccati_ipm as ripm, condensed_ipm as cipm
from qp_cases import CASES, make_case
from helpers import Poly

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def info_of(locp):
    import ctypes as C
    from sofacontrol_amd import _lib
    en, po, dg = C.c_int(), C.c_int(), C.c_int()
    _lib.check(_lib.lib().slocp_condensed_info(C.byref(locp._prob), C.byref(en), C.byref(po), C.byref(dg)), 'slocp_condensed_info')
    return en.value, po.value, dg.value


def solve_product(case, expect):
    from sofacontrol_amd.scp.locp import LOCP
    pl = lambda t: None if t is None else Poly(*t)
    locp = LOCP(case['N'], case['H'], case['Qz'], case['R'], Qzf=case.get('Qzf'), U=pl(case.get('U')), X=pl(case.get('X')),
                Xf=pl(case.get('Xf')), x_char=1. / case['x_scale'])
    assert info_of(locp) == expect, info_of(locp)
    locp.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], case['delta'], case['omega'],
                z=case['z'], zf=case.get('zf'), u=case.get('u_des'))
    J, ok, stats = locp.solve()
    assert ok
    x, u, s = locp.get_solution()
    return x, u, s, J, stats.num_iters


def oracles(case):
    kw = dict(case)
    args = [kw.pop(k) for k in ('N', 'H', 'Qz', 'R', 'Ad', 'Bd', 'dd', 'x0', 'xk', 'delta', 'omega')]
    qp = olocp.build_qp(*args, **kw)
    w, _, info = olocp.solve_exact(qp)
    assert info['status'] == 'optimal'
    xc, uc, Jc, ic = cipm.solve(ripm.Problem(*args, **kw), newton='output')
    return olocp.split(qp, w), olocp.objective(qp, w), (xc, uc, Jc, ic)


@pytest.mark.parametrize('name', ['box_X', 'free', 'box_only_tr_loose', 'terminal_cost'])
def test_condensed_kernel_follows_its_numpy_statement(name):
    case, _ = make_case(**CASES[name])
    (xe, ue, se), Je, (xc, uc, Jc, ic) = oracles(case)
    assert ic['status'] == 'optimal' and ic['inside']
    x, u, s, J, iters = solve_product(case, (1, 2, 1))
    assert iters == ic['iters']                                   # the same interior-point iteration
    assert rel(x, xc) <= 1e-7 and rel(u, uc) <= 1e-7              # ... and the same iterates
    assert rel(x, xe) <= 1e-4 and rel(u, ue) <= 1e-4 and abs(J - Je) <= 1e-7 * max(1.0, abs(Je))


def test_condensed_kernel_at_the_bench_shapes():
    """First QP of a BASELINE C2 (Diamond, X rows) and C5 (Trunk) rollout."""
    import workloads as wl
    from oracle import gusto as ogusto, tpwl as otpwl, pod as opod
    from scipy.interpolate import interp1d
    for w in (wl.diamond_c2(), wl.trunk_c5()):
        N, m, r, dt = w['N'], w['m'], w['r'], w['dt']
        model = dict(w['tab'], w_q=1.0, w_v=0.0)
        X = wl.snapshots(w['q_ref'], 2, seed=2)
        x0 = np.concatenate((np.zeros(r), opod.project(w['U'], w['q_ref'], X)[1]))
        xc_, fc_ = otpwl.characteristic_vals(model)
        z = interp1d(w['t'], w['z'], axis=0)(1.3 + dt * np.arange(N + 1))
        xk = otpwl.rollout(model, w['Ad'], w['Bd'], w['dd'], x0, np.zeros((N, m)))
        A_k, B_k, d_k, _ = ogusto.traj_dynamics(model, w['Ad'], w['Bd'], w['dd'], xk)
        case = dict(N=N, H=w['H'], Qz=w['Qz'], R=w['R'], Ad=A_k, Bd=B_k, dd=d_k, x0=x0, xk=xk, delta=1e4, omega=1.0, z=z,
                    U=(w['UA'], w['Ub']), X=(w['XA'], w['Xb']) if w['XA'] is not None else None, x_scale=1. / np.abs(xc_))
        kw = dict(case)
        args = [kw.pop(k) for k in ('N', 'H', 'Qz', 'R', 'Ad', 'Bd', 'dd', 'x0', 'xk', 'delta', 'omega')]
        xc, uc, Jc, ic = cipm.solve(ripm.Problem(*args, **kw), newton='output')
        x, u, s, J, iters = solve_product(case, (1, 2, 1))
        assert ic['status'] == 'optimal' and ic['inside'] and iters == ic['iters']
        assert rel(x, xc) <= 1e-6 and rel(u, uc) <= 1e-6 and abs(J - Jc) <= 1e-9 * abs(Jc)


def test_condensed_kernel_general_blocks():
    """Dense R and coupled input rows (the input Hessian blocks need their own Cholesky factors), state rows in a third
    output direction (po = 3: generic column mixing), a terminal set."""
    case, _ = make_case(r=4, m=3, P=7, N=12, seed=50, use_X=True)
    m = 3
    rng = np.random.default_rng(51)
    case['R'] = 1e-5 * (np.eye(m) + 0.3 * np.ones((m, m)))
    UA, Ub = case['U']
    case['U'] = (np.vstack([UA, np.ones((1, m))]), np.concatenate([Ub, [1500.0]]))          # + a coupled row
    XA, Xb = case['X']
    extra = case['H'][5:6]                                                                       # the tip z row
    case['X'] = (np.vstack([XA, extra, -extra]), np.concatenate([Xb, [0.05, 0.05]]))
    case['Xf'] = (extra, np.array([0.04]))
    (xe, ue, se), Je, (xc, uc, Jc, ic) = oracles(case)
    assert ic['status'] == 'optimal' and ic['inside']
    x, u, s, J, iters = solve_product(case, (1, 3, 0))
    assert iters == ic['iters'] and rel(x, xc) <= 1e-7 and rel(u, uc) <= 1e-7
    assert rel(x, xe) <= 1e-4 and rel(u, ue) <= 1e-4 and abs(J - Je) <= 1e-7 * max(1.0, abs(Je))


@pytest.mark.parametrize('name', ['tr_active_small_delta', 'tr_active_big_omega', 'warm_centre'])
def test_minimiser_outside_the_trust_region_goes_to_the_full_solve(name):
    """Condensed pass first, then the stage-wise Riccati solve of the full QP: the result is the full QP's."""
    case, _ = make_case(**CASES[name])
    (xe, ue, se), Je, (xc, uc, Jc, ic) = oracles(case)
    assert not ic['inside']
    x, u, s, J, _ = solve_product(case, (1, 2, 1))
    assert rel(x, xe) <= 1e-4 and rel(u, ue) <= 1e-4 and abs(J - Je) <= 1e-7 * max(1.0, abs(Je))


def test_riccati_prescreen_path_without_the_condensed_one():
    """SRH_QP_NO_COND=1 (read when the problem constants are built) keeps the round-1 path: same answers."""
    os.environ['SRH_QP_NO_COND'] = '1'
    try:
        for name in ('box_X', 'terminal_cost'):
            case, _ = make_case(**CASES[name])
            (xe, ue, se), Je, _ = oracles(case)
            x, u, s, J, _ = solve_product(case, (0, 0, 1))       # (enabled, outputs, diagonal input blocks)
            assert rel(x, xe) <= 1e-4 and rel(u, ue) <= 1e-4 and abs(J - Je) <= 1e-7 * max(1.0, abs(Je))
    finally:
        del os.environ['SRH_QP_NO_COND']
