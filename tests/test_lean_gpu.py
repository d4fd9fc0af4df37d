"""GPU: the lean condensed kernels (csrc/lean.hip, locp_lean.h: packed G resident in LDS, own kernel without the inlined
Riccati solver) against the fused kernels they replace on the hot path and against the numpy statement of the algorithm.

This file is a CONSISTENCY test (HIP path vs HIP path: two kernel families must agree); the parity tests of the same kernels against
the oracle -- the restated reference loop -- are tests/test_gusto_bench_shapes_gpu.py (GuSTO at the bench shapes, the reference drivers'
horizons, the one-wave form, forced hand-over and cold retry) and tests/test_locp_gpu.py / test_locp_cond_gpu.py (the QP).

* one LOCP QP (sofacontrol/scp/locp.py:218-342) at the C2 / C5 stage shapes through `LOCP` (slocp_solve): lean vs fused
  (SRH_LOCP_NO_LEAN=1) -- interior-point iteration counts within one of each other, iterates to 1e-8 -- and vs the numpy condensed statement;
* a trust-region-active QP: the lean kernel hands it over (status LEAN_PENDING) and the fused kernel finishes it;
* the GuSTO loop (gusto.py:283-487) on C2 / C5 rollouts: lean + hand-over vs fused only (SRH_GUSTO_NO_LEAN=1): identical SCP
  iteration counts and (J, delta, omega) traces, trajectories to 2e-6 (the lean kernel warm-starts its QPs, the fused one does
  not), for the capped and the uncapped (500) solve."""
import os

import numpy as np
import pytest
from scipy.interpolate import interp1d

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


class env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def first_qp(w, b=0, B=6, seed=2):
    import workloads as wl
    from oracle import gusto as ogusto, tpwl as otpwl, pod as opod
    N, m, r, dt = w['N'], w['m'], w['r'], w['dt']
    model = dict(w['tab'], w_q=1.0, w_v=0.0)
    X = wl.snapshots(w['q_ref'], B, seed=seed)
    x0 = np.concatenate((np.zeros((B, r)), opod.project(w['U'], w['q_ref'], X)), axis=1)
    xc, fc = otpwl.characteristic_vals(model)
    zi = interp1d(w['t'], w['z'], axis=0, bounds_error=False, fill_value=(w['z'][0], w['z'][-1]))
    z = zi(b * 10.0 / B + dt * np.arange(N + 1))
    xk = otpwl.rollout(model, w['Ad'], w['Bd'], w['dd'], x0[b], np.zeros((N, m)))
    A_k, B_k, d_k, idx = ogusto.traj_dynamics(model, w['Ad'], w['Bd'], w['dd'], xk)
    return dict(A=A_k, B=B_k, d=d_k, x0=x0[b], xk=xk, z=z, xc=xc)


def locp_solve(w, qp, delta, lean):
    from helpers import Poly
    from sofacontrol_amd.scp.locp import LOCP
    with env(SRH_LOCP_NO_LEAN=None if lean else '1'):
        locp = LOCP(w['N'], w['H'], w['Qz'], w['R'], U=Poly(w['UA'], w['Ub']),
                    X=Poly(w['XA'], w['Xb']) if w['XA'] is not None else None, x_char=qp['xc'])
        locp.update(list(qp['A']), list(qp['B']), list(qp['d']), qp['x0'], qp['xk'], delta, 1.0, z=qp['z'])
        J, ok, st = locp.solve()
        x, u, s = locp.get_solution()
    return J, ok, st.num_iters, x, u


@pytest.mark.parametrize('which', ['c2', 'c5'])
def test_lean_qp_matches_fused_kernel_and_numpy_statement(which):
    import workloads as wl
    from oracle import condensed_ipm as cipm, riccati_ipm as ripm
    w = wl.diamond_c2() if which == 'c2' else wl.trunk_c5()
    for b in (0, 3):
        qp = first_qp(w, b=b)
        Jl, okl, itl, xl, ul = locp_solve(w, qp, 1e4, lean=True)
        Jf, okf, itf, xf, uf = locp_solve(w, qp, 1e4, lean=False)
        # (the last interior-point iteration is a rounding matter: the stopping test rd <= 1e-9 sd is met at 1e-9-ish either way)
        assert okl and okf and abs(itl - itf) <= 1, (itl, itf)
        assert rel(xl, xf) <= 1e-8 and rel(ul, uf) <= 1e-8 and abs(Jl - Jf) <= 1e-9 * abs(Jf), (rel(xl, xf), rel(ul, uf))
        p = ripm.Problem(w['N'], w['H'], w['Qz'], w['R'], qp['A'], qp['B'], qp['d'], qp['x0'], qp['xk'], 1e4, 1.0, z=qp['z'],
                         U=(w['UA'], w['Ub']), X=(w['XA'], w['Xb']) if w['XA'] is not None else None, x_scale=1.0 / np.abs(qp['xc']))
        xe, ue, Je, inf = cipm.solve(p)
        assert inf['status'] == 'optimal' and inf['inside']
        assert rel(xl, xe) <= 1e-7 and rel(ul, ue) <= 1e-7, (rel(xl, xe), rel(ul, ue))


def test_lean_qp_hands_trust_region_active_qp_to_the_fused_kernel():
    import workloads as wl
    w = wl.diamond_c2()
    qp = first_qp(w, b=0)
    Jl, okl, itl, xl, ul = locp_solve(w, qp, 0.1, lean=True)
    Jf, okf, itf, xf, uf = locp_solve(w, qp, 0.1, lean=False)
    assert okl and okf and itl == itf
    # the minimiser of the QP without its trust-region rows leaves the trust region (numpy statement): the lean kernel cannot
    # accept it, the fused kernel solved the full QP in both runs
    from oracle import condensed_ipm as cipm, riccati_ipm as ripm
    p = ripm.Problem(w['N'], w['H'], w['Qz'], w['R'], qp['A'], qp['B'], qp['d'], qp['x0'], qp['xk'], 0.1, 1.0, z=qp['z'],
                     U=(w['UA'], w['Ub']), X=(w['XA'], w['Xb']), x_scale=1.0 / np.abs(qp['xc']))
    assert not cipm.solve(p)[3]['inside']
    assert rel(xl, xf) <= 1e-12 and rel(ul, uf) <= 1e-12 and abs(Jl - Jf) <= 1e-12 * abs(Jf)


def gusto_case(w, tip_node, B, seed, lean, cap):
    from test_gusto_bench_shapes_gpu import problem
    from sofacontrol_amd.scp.gusto import GuSTO
    from sofacontrol_amd.utils import Polyhedron
    gm, xc, fc, x0, u_init, x_init, z = problem(w, B, seed, tip_node)
    X = Polyhedron(w['XA'], w['Xb']) if w['XA'] is not None else None
    with env(SRH_GUSTO_NO_LEAN=None if lean else '1'):
        g = GuSTO(gm, w['N'], w['dt'], w['Qz'], w['R'], x0, u_init, x_init, z=z, U=Polyhedron(w['UA'], w['Ub']), X=X,
                  x_char=xc, f_char=fc, convg_thresh=1e-3, batch=B, max_trace=512, max_gusto_iters=cap)
        first = (g.iters.copy(), g.status.copy(), g.trace.copy(), g.xopt.copy(), g.uopt.copy())
        g.solve_batch(x0, u_init, x_init, z=z)
        second = (g.iters.copy(), g.status.copy(), g.trace.copy(), g.xopt.copy(), g.uopt.copy())
    return first, second


@pytest.mark.parametrize('which,B', [('c2', 12), ('c5', 8)])
def test_lean_gusto_matches_fused_gusto(which, B):
    import workloads as wl
    w = wl.diamond_c2() if which == 'c2' else wl.trunk_c5()
    tip = 1354 if which == 'c2' else w['tip_node']
    lean = gusto_case(w, tip, B, 2 if which == 'c2' else 9, True, 5)
    fused = gusto_case(w, tip, B, 2 if which == 'c2' else 9, False, 5)
    for (il, sl, tl, xl, ul), (i_f, sf, tf, xf, uf), what in zip(lean, fused, ('uncapped constructor solve', 'capped solve')):
        assert (il == i_f).all(), (what, il, i_f)
        assert (sl == sf).all(), (what, sl, sf)
        for b in range(B):
            k = int(il[b])
            np.testing.assert_allclose(tl[b, :k, :3], tf[b, :k, :3], rtol=1e-7, err_msg='%s rollout %d' % (what, b))
        # (round 4: the lean kernel starts every QP after the first from the previous one's minimiser and multipliers, the fused
        # kernel starts cold -- both stop on the same rule at the same tolerances and agree as far as the flat QP lets two exact
        # solvers agree: 4e-7 measured at the Trunk shape, DESIGN.md section 5 "QP conditioning")
        assert rel(xl, xf) <= 2e-6 and rel(ul, uf) <= 2e-6, (what, rel(xl, xf), rel(ul, uf))
