"""CPU: pin the QP oracle.  The reference's own solver (cvxpy -> OSQP) is absent (parity unpinned,
see oracle/locp.py); the QP is pinned by (i) closed-form dense KKT when no inequality is present,
(ii) KKT-residual certificates of the exact solution, (iii) two independent exact solvers (sparse
generic IPM vs stage-structured Riccati IPM), (iv) scipy's trust-constr on a tiny instance, and (v) a
restatement of OSQP's published algorithm at cvxpy's default tolerance."""
import numpy as np
import pytest

from oracle import locp as olocp, riccati_ipm as ripm
from qp_cases import CASES, make_case


def build(case):
    return olocp.build_qp(case['N'], case['H'], case['Qz'], case['R'], case['Ad'], case['Bd'], case['dd'],
                          case['x0'], case['xk'], case['delta'], case['omega'], z=case['z'],
                          Qzf=case.get('Qzf'), zf=case.get('zf'), U=case['U'], X=case['X'],
                          x_scale=case['x_scale'])


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


@pytest.mark.parametrize('name', list(CASES))
def test_exact_solvers_agree_and_certify(name):
    case, _ = make_case(**CASES[name])
    qp = build(case)
    w, (y, lam), info = olocp.solve_exact(qp, tol=1e-12)
    assert info.get('status', 'optimal') == 'optimal'
    cert = olocp.kkt_certificate(qp, w, y, lam)
    scale = max(1.0, case['omega'])
    assert cert['stationarity'] <= 1e-7 * scale
    assert cert['equality'] <= 1e-8
    assert cert['ineq_violation'] <= 1e-8 and cert['dual_negativity'] == 0.0
    assert cert['complementarity'] <= 1e-7 * scale
    xe, ue, se = olocp.split(qp, w)
    x, u, s, J, info2 = ripm.solve(ripm.Problem(**case), tol=1e-12)
    assert info2['status'] == 'optimal'
    # north-star tolerance: <= 1e-4 relative trajectory error (R = 1e-5 leaves u weakly determined)
    assert rel(x, xe) <= 1e-4 and rel(u, ue) <= 1e-4
    assert abs(J - olocp.objective(qp, w)) <= 1e-7 * max(1.0, abs(J))
    if name.startswith('tr_active') or name.startswith('tr_tiny'):
        assert se.max() > 1e-6 or np.max(np.abs(case['x_scale'] * (xe - case['xk']))) >= case['delta'] * 0.999
        np.testing.assert_allclose(s, se, rtol=0, atol=1e-4 * max(1.0, se.max()))


def test_equality_only_closed_form():
    case, _ = make_case(**CASES['free'])
    qp = build(case)
    w, y = olocp.solve_eq_only(qp)
    xe, ue, _ = olocp.split(qp, w)
    # trust region is far from active (delta = 1e4): the full solvers must return the same point
    w2, _, _ = olocp.solve_exact(qp, tol=1e-12)
    x2, u2, _ = olocp.split(qp, w2)
    assert rel(x2, xe) <= 1e-5 and rel(u2, ue) <= 1e-5
    x3, u3, s3, J3, _ = ripm.solve(ripm.Problem(**case), tol=1e-12)
    assert rel(x3, xe) <= 1e-5 and rel(u3, ue) <= 1e-5


def test_tiny_instance_against_scipy_trust_constr():
    from scipy.optimize import minimize, LinearConstraint
    case, _ = make_case(r=2, m=2, P=3, N=4, seed=40, use_X=False, delta=5e-3, omega=50.0)
    qp = build(case)
    P2 = (2.0 * qp.Pq).toarray()
    f = lambda w: 0.5 * w @ P2 @ w + qp.c @ w
    g = lambda w: P2 @ w + qp.c
    cons = [LinearConstraint(qp.E.toarray(), qp.e, qp.e),
            LinearConstraint(qp.G.toarray(), -np.inf, qp.h)]
    w0, _, _ = olocp.solve_exact(qp, tol=1e-12)
    res = minimize(f, np.zeros_like(w0), jac=g, hess=lambda w: P2, constraints=cons, method='trust-constr',
                   options=dict(gtol=1e-10, xtol=1e-12, maxiter=3000))
    # scipy's barrier method stops at barrier_tolerance ~1e-6: the exact solution must be at least as
    # good and within 1e-4 of it
    assert f(w0) <= f(res.x) + 1e-9 and abs(f(res.x) - f(w0)) <= 1e-4 * max(1.0, abs(f(w0)))
    xe, ue, _ = olocp.split(qp, w0)
    xs_, us_, _ = olocp.split(qp, res.x)
    assert rel(xs_, xe) <= 1e-3


def test_osqp_restatement_reaches_reference_accuracy():
    """What the reference's default solver delivers.  R = 1e-5 against Qz = 100 makes the QP nearly flat
    along many input directions: ADMM at cvxpy's default eps = 1e-5 matches the optimal COST to <1 %
    while the trajectory is still far off; at eps = 1e-7 the trajectory agrees to a few %.  The product
    is held to the exact solution (<= 1e-4), which is tighter than what the reference itself computes."""
    case, _ = make_case(**CASES['box_only_tr_loose'])
    qp = build(case)
    w, _, _ = olocp.solve_exact(qp, tol=1e-12)
    Je = olocp.objective(qp, w)
    wo, _, info = olocp.solve_osqp(qp)
    assert info['status'] == 'solved'
    assert 0.0 <= olocp.objective(qp, wo) - Je <= 1e-2 * abs(Je)
    wo, _, info = olocp.solve_osqp(qp, eps_abs=1e-7, eps_rel=1e-7)
    assert info['status'] == 'solved'
    xe, ue, _ = olocp.split(qp, w)
    xo, uo, _ = olocp.split(qp, wo)
    assert rel(xo, xe) <= 5e-2 and rel(uo, ue) <= 5e-2
    assert abs(olocp.objective(qp, wo) - Je) <= 1e-5 * abs(Je)


# ---------------------------------------------------------------- condensed (output-space) interior point
@pytest.mark.parametrize('name', ['box_X', 'free', 'box_only_tr_loose', 'terminal_cost'])
@pytest.mark.parametrize('newton', ['primal', 'output'])
def test_condensed_ipm_matches_exact_solver(name, newton):
    """oracle.condensed_ipm (numpy statement of csrc/locp_cond.h: states eliminated, Newton systems in input / output
    space) against the generic sparse solver on the same QP, for the cases whose trust region is inactive."""
    from oracle import condensed_ipm as cipm
    from qp_cases import make_case, CASES
    case, _ = make_case(**CASES[name])
    kw = dict(case)
    args = [kw.pop(k) for k in ('N', 'H', 'Qz', 'R', 'Ad', 'Bd', 'dd', 'x0', 'xk', 'delta', 'omega')]
    qp = olocp.build_qp(*args, **kw)
    w, _, _ = olocp.solve_exact(qp)
    xe, ue, se = olocp.split(qp, w)
    sp = ripm.Problem(*args, **kw)
    x, u, J, info = cipm.solve(sp, newton=newton)
    assert info['status'] == 'optimal' and info['inside']
    assert np.abs(x - xe).max() <= 1e-6 * max(1e-12, np.abs(xe).max())
    assert np.abs(u - ue).max() <= 1e-6 * max(1e-12, np.abs(ue).max())
    assert abs(J + case['omega'] * se[0] - olocp.objective(qp, w)) <= 1e-8 * abs(olocp.objective(qp, w))


def test_condensed_ipm_reports_a_minimiser_outside_the_trust_region():
    from oracle import condensed_ipm as cipm
    from qp_cases import make_case, CASES
    case, _ = make_case(**CASES['tr_active_small_delta'])
    kw = dict(case)
    args = [kw.pop(k) for k in ('N', 'H', 'Qz', 'R', 'Ad', 'Bd', 'dd', 'x0', 'xk', 'delta', 'omega')]
    x, u, J, info = cipm.solve(ripm.Problem(*args, **kw))
    assert info['status'] == 'optimal' and not info['inside']      # -> the full stage-wise solve takes over


@pytest.mark.parametrize('name', ['box_X', 'box_only_tr_loose', 'terminal_cost'])
def test_condensed_ipm_direction_from_the_solved_system(name):
    """Round 4: dy = G du is read off the solved output-space system (dy_k = Ls_k^-T w_k) instead of a second product with G
    (condensed_ipm.direction_y; kernels: ql::newton_back, qpc::newton_solve; twin: direction_y).  Both forms are the same Newton
    direction: same minimiser, iteration counts within two of each other (the product form passes the K-solve error through K)."""
    from oracle import condensed_ipm as cipm
    case, _ = make_case(**CASES[name])
    kw = dict(case)
    args = [kw.pop(k) for k in ('N', 'H', 'Qz', 'R', 'Ad', 'Bd', 'dd', 'x0', 'xk', 'delta', 'omega')]
    sp = ripm.Problem(*args, **kw)
    xa, ua, Ja, ia = cipm.solve(sp)
    xb, ub, Jb, ib = cipm.solve(sp, dy_from_system=False)
    assert ia['status'] == ib['status'] == 'optimal'
    assert abs(ia['iters'] - ib['iters']) <= 2
    assert np.abs(xa - xb).max() <= 1e-8 * np.abs(xb).max() and np.abs(ua - ub).max() <= 1e-8 * max(1.0, np.abs(ub).max())
    assert abs(Ja - Jb) <= 1e-10 * abs(Jb)


def _problem(name, **over):
    case, extra = make_case(**dict(CASES[name], **over))
    kw = dict(case)
    args = [kw.pop(k) for k in ('N', 'H', 'Qz', 'R', 'Ad', 'Bd', 'dd', 'x0', 'xk', 'delta', 'omega')]
    return ripm.Problem(*args, **kw)


def test_condensed_ipm_warm_start_same_minimiser_fewer_iterations():
    """Round 4 (condensed_ipm.solve(warm=), the rule the lean kernels and the CPU twin follow): a QP started from the minimiser and
    multipliers of a neighbouring QP -- here the same horizon with a slightly different target amplitude, as between two SCP
    iterations -- reaches the same minimiser in fewer interior-point iterations; info['warm'] says which start produced the result."""
    from oracle import condensed_ipm as cipm
    xa, ua, Ja, ia = cipm.solve(_problem('box_X'))
    pb = _problem('box_X', amp=0.16)
    xc, uc, Jc, ic = cipm.solve(pb)
    xw, uw, Jw, iw = cipm.solve(pb, warm=ia['final'])
    assert ia['status'] == ic['status'] == iw['status'] == 'optimal' and iw['warm'] and not ic['warm']
    assert iw['iters'] < ic['iters']
    assert np.abs(xw - xc).max() <= 1e-7 * np.abs(xc).max() and np.abs(uw - uc).max() <= 1e-7 * max(1.0, np.abs(uc).max())
    assert abs(Jw - Jc) <= 1e-9 * abs(Jc)


def test_condensed_ipm_warm_start_that_fails_is_repeated_cold():
    """A warm start the interior point cannot use (non-finite multipliers: the iteration stops as 'failed') must not be the answer:
    the solve is repeated from Mehrotra's point and says so."""
    from oracle import condensed_ipm as cipm
    p = _problem('box_X')
    xc, uc, Jc, ic = cipm.solve(p)
    bad = dict(u=ic['final']['u'].copy(), lx=[None] + [np.full_like(l, np.nan) for l in ic['final']['lx'][1:]],
               lu=[np.full_like(l, np.nan) for l in ic['final']['lu']])
    xw, uw, Jw, iw = cipm.solve(p, warm=bad)
    assert iw['status'] == 'optimal' and not iw['warm'] and iw['iters'] == ic['iters']
    assert np.array_equal(xw, xc) and np.array_equal(uw, uc)
