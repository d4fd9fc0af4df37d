"""GPU parity of the GuSTO kernels at the shapes bench.py times: BASELINE config C2 (Diamond r = 30, n_u = 4, P = 64,
N = 50, U box + X box, figure-8: lean instantiation <4, 60, 4, 50, 7, 4>, fused <false, 4, 60> for what it hands over) and
C5 (Trunk r = 30, n_u = 8, U box: lean <8, 60, 1, 50, 24, 0>, fused <false, 8, 60>), against the restated reference loop
(oracle.gusto, sofacontrol/scp/gusto.py:283-487) around the stage-structured oracle QP.  Same iteration counts,
(J, delta, omega) trace to 1e-6, trajectories <= 1e-4 relative.  Every case asserts WHICH instantiation answered
(`GuSTO.kernel_info`, sgusto_plan_info): the run-time-horizon lean kernels (N = 20), the fixed layout of the shipped
r = 36 Diamond basis and the fixed-vs-run-time pair at C2 each have a case that is guaranteed to reach them."""
import numpy as np
import pytest
from scipy.interpolate import interp1d

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def problem(w, B, seed, tip_node, phase_span=10.0):
    """Product model + B rollouts with different initial states (projected snapshots) and target phases -- the same
    construction as bench.py main() / scp_c5()."""
    import bench
    import workloads as wl
    from sofacontrol_amd.mor.pod import POD
    N, m, r, dt = w['N'], w['m'], w['r'], w['dt']
    rom = POD(dict(U=w['U'], q_ref=w['q_ref'], v_ref=w['v_ref']))
    tp, gm = bench.build_model(w, tip_node)
    xc, fc = gm.get_characteristic_vals()
    X = wl.snapshots(w['q_ref'], B, seed=seed)
    x0 = np.concatenate((np.zeros((B, r)), rom.compute_RO_state(qf=X)), axis=1)
    u_init = np.zeros((B, N, m))
    x_init, _ = tp.rollout(x0, u_init, dt)
    zi = interp1d(w['t'], w['z'], axis=0, bounds_error=False, fill_value=(w['z'][0], w['z'][-1]))
    z = np.stack([zi(b * phase_span / B + dt * np.arange(N + 1)) for b in range(B)])
    return gm, xc, fc, x0, u_init, x_init, z


def oracle_solve(w, xc, fc, x0, u_init, x_init, z, max_iters):
    from oracle import gusto as ogusto
    model = dict(w['tab'], w_q=1.0, w_v=0.0)
    X = (w['XA'], w['Xb']) if w['XA'] is not None else None
    return ogusto.solve(model, w['Ad'], w['Bd'], w['dd'], w['H'], w['N'], w['dt'], w['Qz'], w['R'], x0, u_init, x_init,
                        z=z, U=(w['UA'], w['Ub']), X=X, x_char=xc, f_char=fc, convg_thresh=1e-3,
                        qp_solver='riccati_ipm', max_gusto_iters=max_iters)


def compare(g, b, ref, what):
    xe, ue, ze, tr = ref
    assert int(g.iters[b]) == len(tr), (what, b, int(g.iters[b]), len(tr))
    got = g.trace[b, :len(tr), :3]
    np.testing.assert_allclose(got, np.array([t[:3] for t in tr]), rtol=1e-6, err_msg='%s rollout %d' % (what, b))
    assert rel(g.xopt[b], xe) <= 1e-4 and rel(g.uopt[b], ue) <= 1e-4 and rel(g.zopt[b], ze) <= 1e-4, \
        (what, b, rel(g.xopt[b], xe), rel(g.uopt[b], ue))
    return rel(g.xopt[b], xe), rel(g.uopt[b], ue)


def run_case(w, tip_node, B, seed, variant, what, lean=None, long_solves=2, capped_stays_lean=False):
    from sofacontrol_amd.scp.gusto import GuSTO
    from sofacontrol_amd.utils import Polyhedron
    gm, xc, fc, x0, u_init, x_init, z = problem(w, B, seed, tip_node)
    X = Polyhedron(w['XA'], w['Xb']) if w['XA'] is not None else None
    # the constructor solves with the reference's default cap of 500 SCP iterations (gusto.py:142-147)
    g = GuSTO(gm, w['N'], w['dt'], w['Qz'], w['R'], x0, u_init, x_init, z=z, U=Polyhedron(w['UA'], w['Ub']), X=X,
              x_char=xc, f_char=fc, convg_thresh=1e-3, batch=B, max_trace=512, max_gusto_iters=5)
    assert g._fused and g.variant == variant, g.variant
    info = g.kernel_info
    # the instantiation that answered (what a rocprof trace would show): lean kernel + fused kernel for the hand-overs
    assert info['family'] == 'lean' and info['lean'] == lean and info['fused'] == variant, info
    assert 0 <= info['handed_over'] <= B, info
    assert (g.status == 0).all(), g.status
    worst = [0.0, 0.0]
    for b in range(min(B, long_solves)):            # the long solves: two rollouts
        ex, eu = compare(g, b, oracle_solve(w, xc, fc, x0[b], u_init[b], x_init[b], z[b], 500), what + ' max 500')
        worst = [max(worst[0], ex), max(worst[1], eu)]
    # the cap bench.py uses (its real-time drivers use 0..5): every rollout
    g.solve_batch(x0, u_init, x_init, z=z)
    assert g.iters.max() <= 6
    info = g.kernel_info
    if capped_stays_lean:                           # the capped solves of the bench cases never shrink delta below 312
        assert info['handed_over'] == 0, info
    for b in range(B):
        ex, eu = compare(g, b, oracle_solve(w, xc, fc, x0[b], u_init[b], x_init[b], z[b], 5), what + ' max 5')
        worst = [max(worst[0], ex), max(worst[1], eu)]
    print('%s: %s, worst relative trajectory error x %.2e u %.2e' % (what, info['kernel'], worst[0], worst[1]))
    return g


def test_fused_gusto_diamond_c2_matches_oracle():
    import workloads as wl
    run_case(wl.diamond_c2(), 1354, B=6, seed=2, variant=(False, 4, 60), what='C2', lean=(4, 60, 4, 50, 7, 4), capped_stays_lean=True)


def test_half_size_lean_workgroup_c2_matches_oracle_and_the_full_size_kernel(monkeypatch):
    """The half-size lean workgroup (round 6: 256 threads, <= 80 KB of LDS, every packed row of G in L2, a thread owns an input AND a
    state-row slot, two-pass condensation -- two rollouts per CU; chosen by itself for batches above the CU count, SRH_LEAN_HALF=1 here):
    instantiation <4, 60, 4, 50, 50, 4>.  Same acceptance as the full-size C2 case (uncapped and capped solves against oracle.gusto,
    identical SCP iteration counts and (J, delta, omega) traces), and the same trajectories as the full-size kernel to 1e-8; a
    keep_solver_state solve (warm start of the first QP from the previous solve) and a forced hand-over go through it as well."""
    import workloads as wl
    from sofacontrol_amd.scp.gusto import GuSTO
    from sofacontrol_amd.utils import Polyhedron
    w = wl.diamond_c2()
    monkeypatch.setenv('SRH_LEAN_HALF', '1')
    gh = run_case(w, 1354, B=6, seed=2, variant=(False, 4, 60), what='C2 half-size workgroup', lean=(4, 60, 4, 50, 50, 4), capped_stays_lean=True)
    assert gh.kernel_info['lds_bytes_lean'] <= 80 * 1024
    xh, uh, ih = gh.xopt.copy(), gh.uopt.copy(), gh.iters.copy()
    monkeypatch.setenv('SRH_LEAN_HALF', '0')
    gm, xc, fc, x0, u_init, x_init, z = problem(w, 6, 2, 1354)
    kw = dict(z=z, U=Polyhedron(w['UA'], w['Ub']), X=Polyhedron(w['XA'], w['Xb']), x_char=xc, f_char=fc, convg_thresh=1e-3, batch=6,
              max_trace=0, max_gusto_iters=5, first_solve_cap=5)
    gf = GuSTO(gm, w['N'], w['dt'], w['Qz'], w['R'], x0, u_init, x_init, **kw)
    assert gf.kernel_info['lean'] == (4, 60, 4, 50, 7, 4)
    gf.solve_batch(x0, u_init, x_init, z=z)
    assert (gf.iters == ih).all()
    assert rel(xh, gf.xopt) <= 1e-8 and rel(uh, gf.uopt) <= 1e-8, (rel(xh, gf.xopt), rel(uh, gf.uopt))
    # solver state kept between solves, and a forced hand-over of SCP iteration 1 to the fused kernel (512 threads, its own layout)
    monkeypatch.setenv('SRH_LEAN_HALF', '1')
    gk = GuSTO(gm, w['N'], w['dt'], w['Qz'], w['R'], x0, u_init, x_init, keep_solver_state=True, **kw)
    assert gk.kernel_info['lean'] == (4, 60, 4, 50, 50, 4) and gk.solver_state_kept
    gk.solve_batch(x0, u_init, x_init, z=z)
    gk.solve_batch(x0, u_init, x_init, z=z)
    assert (gk.iters == ih).all() and rel(gk.xopt, xh) <= 1e-7 and rel(gk.uopt, uh) <= 1e-7
    monkeypatch.setenv('SRH_LEAN_FORCE_HANDOVER', '1')
    go = GuSTO(gm, w['N'], w['dt'], w['Qz'], w['R'], x0, u_init, x_init, **kw)
    monkeypatch.delenv('SRH_LEAN_FORCE_HANDOVER', raising=False)
    go.solve_batch(x0, u_init, x_init, z=z)
    assert int(go.kernel_info['handed_over']) == int((ih >= 2).sum()) > 0
    assert (go.iters == ih).all() and rel(go.xopt, xh) <= 1e-6 and rel(go.uopt, uh) <= 1e-6


def test_half_size_lean_workgroup_is_chosen_for_batches_above_the_cu_count():
    """Plan creation picks the layout by batch size: up to 256 rollouts the full-size workgroup (one rollout per CU is faster for each
    of them), above that the half-size one (two per CU: throughput).  300 C2 rollouts, capped solves: both layouts give the same SCP
    iteration counts and trajectories; the first 2 rollouts against the oracle."""
    import workloads as wl
    from sofacontrol_amd.scp.gusto import GuSTO
    from sofacontrol_amd.utils import Polyhedron
    w = wl.diamond_c2()
    B = 300
    gm, xc, fc, x0, u_init, x_init, z = problem(w, B, 2, 1354)
    kw = dict(U=Polyhedron(w['UA'], w['Ub']), X=Polyhedron(w['XA'], w['Xb']), x_char=xc, f_char=fc, convg_thresh=1e-3, max_gusto_iters=5,
              first_solve_cap=5)
    g = GuSTO(gm, w['N'], w['dt'], w['Qz'], w['R'], x0, u_init, x_init, z=z, batch=B, max_trace=16, **kw)
    assert g.kernel_info['lean'] == (4, 60, 4, 50, 50, 4), g.kernel_info
    g2 = GuSTO(gm, w['N'], w['dt'], w['Qz'], w['R'], x0[:200], u_init[:200], x_init[:200], z=z[:200], batch=200, max_trace=0, **kw)
    assert g2.kernel_info['lean'] == (4, 60, 4, 50, 7, 4), g2.kernel_info
    assert (g.status == 0).all() and (g2.status == 0).all()
    assert (g.iters[:200] == g2.iters).all()
    assert rel(g.xopt[:200], g2.xopt) <= 1e-8 and rel(g.uopt[:200], g2.uopt) <= 1e-8
    for b in range(2):
        compare(g, b, oracle_solve(w, xc, fc, x0[b], u_init[b], x_init[b], z[b], 5), 'C2 half-size, batch 300')


def test_fused_gusto_trunk_c5_matches_oracle():
    import workloads as wl
    w = wl.trunk_c5()
    run_case(w, w['tip_node'], B=4, seed=9, variant=(False, 8, 60), what='C5', lean=(8, 60, 1, 50, 24, 0), capped_stays_lean=True)


def test_lean_runtime_horizon_n20_matches_oracle():
    """N = 20 at n_x = 60: no fixed-layout instantiation exists for it -- the run-time-horizon lean kernel
    <4, 60, 4, 0, 0, 0> answers (box rows, 4 state rows)."""
    import workloads as wl
    run_case(wl.diamond_c2(N=20), 1354, B=3, seed=2, variant=(False, 4, 60), what='C2 with N = 20', lean=(4, 60, 4, 0, 0, 0),
             long_solves=1)


def test_lean_fixed_layout_r36_matches_oracle():
    """The reference's SHIPPED Diamond basis size r = 36 (examples/diamond/pod_model.pkl): n_x = 72, N = 50, 4 state rows --
    the fixed-layout instantiation <4, 72, 4, 50, 18, 4>; what it hands over goes to the split-panel fused kernel."""
    import workloads as wl
    run_case(wl.diamond_c2(r=36), 1354, B=3, seed=2, variant=(True, 4, 72), what='C2 at r = 36', lean=(4, 72, 4, 50, 18, 4),
             long_solves=1)


def test_lean_fixed_layout_equals_runtime_layout_at_c2(monkeypatch):
    """SRH_LEAN_NO_FIXED=1 (read when the plan is created) selects the run-time-size lean kernel for the same problem: same
    SCP iteration counts, same status, trajectories to rounding (the two differ in index arithmetic only)."""
    import workloads as wl
    from sofacontrol_amd.scp.gusto import GuSTO
    from sofacontrol_amd.utils import Polyhedron
    w = wl.diamond_c2()
    B = 8
    gm, xc, fc, x0, u_init, x_init, z = problem(w, B, 2, 1354)
    res = {}
    for tag in ('fixed', 'runtime'):
        if tag == 'runtime':
            monkeypatch.setenv('SRH_LEAN_NO_FIXED', '1')
        g = GuSTO(gm, w['N'], w['dt'], w['Qz'], w['R'], x0, u_init, x_init, z=z, U=Polyhedron(w['UA'], w['Ub']),
                  X=Polyhedron(w['XA'], w['Xb']), x_char=xc, f_char=fc, convg_thresh=1e-3, batch=B, max_trace=0, max_gusto_iters=5)
        long_iters = g.iters.copy()
        g.solve_batch(x0, u_init, x_init, z=z)
        res[tag] = (g.kernel_info['lean'], long_iters, g.iters.copy(), g.status.copy(), g.xopt.copy(), g.uopt.copy())
    monkeypatch.delenv('SRH_LEAN_NO_FIXED')
    assert res['fixed'][0] == (4, 60, 4, 50, 7, 4) and res['runtime'][0] == (4, 60, 4, 0, 0, 0), (res['fixed'][0], res['runtime'][0])
    assert (res['fixed'][1] == res['runtime'][1]).all() and (res['fixed'][2] == res['runtime'][2]).all()
    assert (res['fixed'][3] == res['runtime'][3]).all()
    assert rel(res['fixed'][4], res['runtime'][4]) <= 1e-9 and rel(res['fixed'][5], res['runtime'][5]) <= 1e-9


@pytest.mark.parametrize('N,dt,with_X,cap,lean', [(5, 0.05, True, 500, (4, 60, 4, -1, 0, 0)), (3, 0.1, False, 5, (4, 60, 1, -1, 0, 0)),
                                                   (200, 0.05, False, 500, None)])
def test_reference_driver_horizons_match_oracle(N, dt, with_X, cap, lean):
    """The horizons the reference's own Diamond drivers solve (bench.py: secondary.scp_reference_horizons): N = 5 / dt = 0.05 with
    the X box (examples/diamond/diamond.py:309-316), N = 3 / dt = 0.1 capped at 5 SCP iterations (examples/hardware/diamond.py:
    393-399; no state rows) -- the short-horizon lean kernels <4, 60, GX, -1, 0, 0>, interior point on one wave (ql::ipm_wave) -- and the open-loop N = 200 (examples/hardware/diamond.py:471-474): N p_o = 400
    outputs exceed the condensed path's 128, the stage-wise Riccati kernel answers.  Against oracle.gusto at that N."""
    import workloads as wl
    from sofacontrol_amd.scp.gusto import GuSTO
    from sofacontrol_amd.utils import Polyhedron
    w = wl.diamond_c2(N=N, dt=dt, with_X=with_X)
    B = 2 if N == 200 else 4
    gm, xc, fc, x0, u_init, x_init, z = problem(w, B, 2, 1354)
    X = Polyhedron(w['XA'], w['Xb']) if with_X else None
    g = GuSTO(gm, N, dt, w['Qz'], w['R'], x0, u_init, x_init, z=z, U=Polyhedron(w['UA'], w['Ub']), X=X, x_char=xc, f_char=fc,
              convg_thresh=1e-3, batch=B, max_trace=512, max_gusto_iters=cap)
    info = g.kernel_info
    assert info['family'] == ('lean' if lean else 'fused') and info['lean'] == lean, info
    g.solve_batch(x0, u_init, x_init, z=z)
    for b in range(B):
        compare(g, b, oracle_solve(w, xc, fc, x0[b], u_init[b], x_init[b], z[b], cap), 'N = %d' % N)


def test_lean_cold_retry_after_failed_warm_start(monkeypatch):
    """The lean kernel warm-starts every QP after the first of a solve and REPEATS a QP cold when the warm-started interior point
    does not converge (csrc/lean.hip, the `attempt` loop; never taken on the bench batch).  The test knob SRH_LEAN_POISON_WARM=1
    (read when the plan is created -> GustoPar::poison_warm) replaces the warm multipliers by +inf: every warm attempt fails at
    once and the retry answers.  The poisoned plan must (1) give the SCP trace of the numpy statement with every QP cold-started
    (oracle.gusto(..., qp_solver='condensed_ipm', warm_start_qp=False): the same QP sequence), (2) agree with the unpoisoned plan:
    equal SCP iteration counts and status, trajectories to 1e-6 (two exact solves of the same QPs stopped at the same gap)."""
    import workloads as wl
    from oracle import gusto as ogusto
    from sofacontrol_amd.scp.gusto import GuSTO
    from sofacontrol_amd.utils import Polyhedron
    w = wl.diamond_c2()
    B, cap = 4, 5
    gm, xc, fc, x0, u_init, x_init, z = problem(w, B, 2, 1354)
    res = {}
    for tag in ('plain', 'poisoned'):
        if tag == 'poisoned':
            monkeypatch.setenv('SRH_LEAN_POISON_WARM', '1')
        else:
            monkeypatch.delenv('SRH_LEAN_POISON_WARM', raising=False)
        g = GuSTO(gm, w['N'], w['dt'], w['Qz'], w['R'], x0, u_init, x_init, z=z, U=Polyhedron(w['UA'], w['Ub']), X=Polyhedron(w['XA'], w['Xb']),
                  x_char=xc, f_char=fc, convg_thresh=1e-3, batch=B, max_trace=64, max_gusto_iters=cap)
        assert g.kernel_info['family'] == 'lean', g.kernel_info
        g.solve_batch(x0, u_init, x_init, z=z)
        assert int(g.kernel_info['handed_over']) == 0, g.kernel_info          # the retry answered inside the lean kernel
        res[tag] = (g.iters.copy(), g.status.copy(), g.xopt.copy(), g.uopt.copy(), g.trace.copy())
    monkeypatch.delenv('SRH_LEAN_POISON_WARM', raising=False)
    a, p = res['plain'], res['poisoned']
    assert (a[0] == p[0]).all() and (a[1] == p[1]).all(), (a[0], p[0], a[1], p[1])
    assert (p[0] >= 2).all()                                                  # at least one QP per rollout took the retry
    assert rel(p[2], a[2]) <= 1e-6 and rel(p[3], a[3]) <= 1e-6, (rel(p[2], a[2]), rel(p[3], a[3]))
    model = dict(w['tab'], w_q=1.0, w_v=0.0)
    for b in range(2):
        xe, ue, ze, tr = ogusto.solve(model, w['Ad'], w['Bd'], w['dd'], w['H'], w['N'], w['dt'], w['Qz'], w['R'], x0[b], u_init[b], x_init[b],
                                      z=z[b], U=(w['UA'], w['Ub']), X=(w['XA'], w['Xb']), x_char=xc, f_char=fc, convg_thresh=1e-3,
                                      qp_solver='condensed_ipm', warm_start_qp=False, max_gusto_iters=cap)
        assert int(p[0][b]) == len(tr)
        np.testing.assert_allclose(p[4][b, :len(tr), :3], np.array([t[:3] for t in tr]), rtol=1e-6)
        assert rel(p[2][b], xe) <= 1e-6 and rel(p[3][b], ue) <= 1e-6, (rel(p[2][b], xe), rel(p[3][b], ue))


@pytest.mark.parametrize('which,N,dt,with_X', [('diamond', 5, 0.05, True), ('diamond', 3, 0.1, False), ('diamond', 8, 0.05, True), ('trunk', 8, 0.1, False),
                                                ('trunk', 4, 0.1, False), ('diamond', 1, 0.05, True), ('diamond', 2, 0.05, False)])
def test_short_horizon_wave_form_matches_box_form_and_oracle(which, N, dt, with_X, monkeypatch):
    """Short horizons (N p_o <= 16: K is one tile) run the interior point on ONE wave (ql::ipm_wave, lean<M, NX, GX, -1, 0, 0>); with
    SRH_LEAN_NO_WAVE=1 at plan creation the same problem takes the eight-wave form (ql::ipm_box, run-time horizon).  Both are the same
    iteration: equal SCP iteration counts and status, trajectories to 1e-7; and the one-wave form against oracle.gusto.  Diamond
    (n_u = 4, with / without the X box) and Trunk (n_u = 8: N = 8 fills all 64 lanes)."""
    import workloads as wl
    from sofacontrol_amd.scp.gusto import GuSTO
    from sofacontrol_amd.utils import Polyhedron
    if which == 'diamond':
        w, tip, seed = wl.diamond_c2(N=N, dt=dt, with_X=with_X), 1354, 2
    else:
        w = wl.trunk_c5(N=N, dt=dt)
        tip, seed = w['tip_node'], 9
    B, cap = 4, 5
    gm, xc, fc, x0, u_init, x_init, z = problem(w, B, seed, tip)
    X = Polyhedron(w['XA'], w['Xb']) if (with_X and w['XA'] is not None) else None
    res = {}
    for tag in ('wave', 'box'):
        if tag == 'box':
            monkeypatch.setenv('SRH_LEAN_NO_WAVE', '1')
        else:
            monkeypatch.delenv('SRH_LEAN_NO_WAVE', raising=False)
        g = GuSTO(gm, N, dt, w['Qz'], w['R'], x0, u_init, x_init, z=z, U=Polyhedron(w['UA'], w['Ub']), X=X, x_char=xc, f_char=fc,
                  convg_thresh=1e-3, batch=B, max_trace=64, max_gusto_iters=cap)
        info = g.kernel_info
        assert info['family'] == 'lean' and (info['lean'][3] == -1) == (tag == 'wave'), info
        g.solve_batch(x0, u_init, x_init, z=z)
        res[tag] = (g, g.iters.copy(), g.status.copy(), g.xopt.copy(), g.uopt.copy())
    monkeypatch.delenv('SRH_LEAN_NO_WAVE', raising=False)
    a, b = res['wave'], res['box']
    assert (a[1] == b[1]).all() and (a[2] == b[2]).all(), (a[1], b[1], a[2], b[2])
    assert rel(a[3], b[3]) <= 1e-7 and rel(a[4], b[4]) <= 1e-7, (rel(a[3], b[3]), rel(a[4], b[4]))
    for bi in range(2):
        compare(a[0], bi, oracle_solve(w, xc, fc, x0[bi], u_init[bi], x_init[bi], z[bi], cap), '%s N = %d one-wave' % (which, N))


@pytest.mark.parametrize('N,batch,zero_copy', [(5, 2, True), (5, 2, False), (50, 2, True), (50, 40, False)])
def test_forced_hand_over_gives_the_same_solve(N, batch, zero_copy, monkeypatch):
    """The hand-over protocol end to end, deterministically: SRH_LEAN_FORCE_HANDOVER=1 (read at plan creation) makes the lean kernel
    hand SCP iteration 1 of EVERY rollout to the fused kernel as 'minimiser outside the trust region' (resume record -> fused kernel
    in resume mode -> full QP on the Riccati path).  The full QP has the same minimiser, so iteration counts, status and trajectories
    must equal the unforced solve's (1e-6).  Through both host paths of sgusto_plan_solve: zero-copy (small batches: the lean
    launch alone, the host reads the status words and launches the fused kernel only then) and the copying form
    (SRH_GUSTO_NO_ZEROCOPY=1 / batches above 1 MiB of arguments: hand-over counter + unconditional resume launch)."""
    import workloads as wl
    from sofacontrol_amd.scp.gusto import GuSTO
    from sofacontrol_amd.utils import Polyhedron
    w = wl.diamond_c2(N=N, dt=0.05)
    gm, xc, fc, x0, u_init, x_init, z = problem(w, batch, 2, 1354)
    if not zero_copy and batch < 32:
        monkeypatch.setenv('SRH_GUSTO_NO_ZEROCOPY', '1')
    res = {}
    for tag in ('plain', 'forced'):
        if tag == 'forced':
            monkeypatch.setenv('SRH_LEAN_FORCE_HANDOVER', '1')
        else:
            monkeypatch.delenv('SRH_LEAN_FORCE_HANDOVER', raising=False)
        g = GuSTO(gm, N, 0.05, w['Qz'], w['R'], x0, u_init, x_init, z=z, U=Polyhedron(w['UA'], w['Ub']), X=Polyhedron(w['XA'], w['Xb']),
                  x_char=xc, f_char=fc, convg_thresh=1e-3, batch=batch, max_trace=16, max_gusto_iters=5)
        assert g.kernel_info['family'] == 'lean'
        g.solve_batch(x0, u_init, x_init, z=z)
        ho = int(g.kernel_info['handed_over'])
        res[tag] = (g.iters.copy(), g.status.copy(), g.xopt.copy(), g.uopt.copy(), ho)
    monkeypatch.delenv('SRH_LEAN_FORCE_HANDOVER', raising=False)
    a, f = res['plain'], res['forced']
    assert a[4] == 0, a[4]
    assert f[4] == int((a[0] >= 2).sum()), (f[4], a[0])          # every rollout that reaches SCP iteration 1 was handed over
    assert f[4] > 0
    assert (a[0] == f[0]).all() and (a[1] == f[1]).all(), (a[0], f[0], a[1], f[1])
    assert rel(f[2], a[2]) <= 1e-6 and rel(f[3], a[3]) <= 1e-6, (rel(f[2], a[2]), rel(f[3], a[3]))


def test_handed_over_zero_copy_solve_is_not_slower_than_the_copying_form(monkeypatch):
    """A zero-copy solve passes pinned HOST pointers to the kernels (gusto.hip).  The N = 50 lean kernels and the fused kernel used to
    read x0 / z / zf / u_des across PCIe in every QP and interior-point iteration -- unbounded for exactly the slow solves (round-5
    advice); they now work on copies in the work block (GustoBatch::host_args).  One C2 rollout, SCP iteration 1 forced to the fused
    kernel (Riccati path, ~20 interior-point iterations over the target): the zero-copy form must not take longer than the copying
    form (generous 1.3x for timer noise; it should be faster by the saved runtime calls) and must return the same solve."""
    import time
    import workloads as wl
    from sofacontrol_amd.scp.gusto import GuSTO
    from sofacontrol_amd.utils import Polyhedron
    w = wl.diamond_c2(N=50, dt=0.05)
    gm, xc, fc, x0, u_init, x_init, z = problem(w, 1, 2, 1354)
    monkeypatch.setenv('SRH_LEAN_FORCE_HANDOVER', '1')
    g = GuSTO(gm, 50, 0.05, w['Qz'], w['R'], x0, u_init, x_init, z=z, U=Polyhedron(w['UA'], w['Ub']), X=Polyhedron(w['XA'], w['Xb']),
              x_char=xc, f_char=fc, convg_thresh=1e-3, batch=1, max_trace=0, max_gusto_iters=5)
    monkeypatch.delenv('SRH_LEAN_FORCE_HANDOVER', raising=False)
    out, ms = {}, {}
    for tag in ('zero_copy', 'copying', 'zero_copy', 'copying'):
        if tag == 'copying':
            monkeypatch.setenv('SRH_GUSTO_NO_ZEROCOPY', '1')
        else:
            monkeypatch.delenv('SRH_GUSTO_NO_ZEROCOPY', raising=False)
        g.solve_batch(x0, u_init, x_init, z=z)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            g.solve_batch(x0, u_init, x_init, z=z)
            ts.append(time.perf_counter() - t0)
        ms[tag] = min(ts) * 1e3
        assert int(g.kernel_info['handed_over']) == 1
        out[tag] = (g.iters.copy(), g.xopt.copy(), g.uopt.copy())
    monkeypatch.delenv('SRH_GUSTO_NO_ZEROCOPY', raising=False)
    print('handed-over C2 solve: zero-copy %.3f ms, copying %.3f ms' % (ms['zero_copy'], ms['copying']))
    assert (out['zero_copy'][0] == out['copying'][0]).all()
    assert rel(out['zero_copy'][1], out['copying'][1]) <= 1e-9 and rel(out['zero_copy'][2], out['copying'][2]) <= 1e-9
    assert ms['zero_copy'] <= 1.3 * ms['copying'], ms


@pytest.mark.parametrize('delta0', [1.0, 4.0])
def test_binding_trust_region_qps_follow_the_oracle(delta0, monkeypatch):
    """QPs whose trust region BINDS, chosen rather than waited for: with delta0 = 1 or 4 (instead of the reference's 1e4, gusto.py:142-147)
    the minimiser of the QP without its trust-region rows leaves the region, the lean kernel hands the rollout to the fused kernel and
    the full QP (trust-region rows active: the stage-wise Riccati interior point) is solved from there on.  delta0 = 1: the step stays
    outside the region by more than epsilon in every iteration (omega x 5 each time: gusto.py:383-402); delta0 = 4: binding QPs whose
    steps are accepted, delta halved on the way.  The numpy statement takes the same route (condensed attempt, then oracle.riccati_ipm)
    -- its calls are counted: at least five binding QPs -- and the GPU's (J, delta, omega) trace must follow it to 1e-6, the
    trajectories to 1e-4."""
    import workloads as wl
    from oracle import gusto as ogusto, riccati_ipm as ripm
    from sofacontrol_amd.scp.gusto import GuSTO
    from sofacontrol_amd.utils import Polyhedron
    w = wl.diamond_c2()
    B, cap = 2, 6
    gm, xc, fc, x0, u_init, x_init, z = problem(w, B, 2, 1354)
    g = GuSTO(gm, w['N'], w['dt'], w['Qz'], w['R'], x0, u_init, x_init, z=z, U=Polyhedron(w['UA'], w['Ub']), X=Polyhedron(w['XA'], w['Xb']),
              x_char=xc, f_char=fc, convg_thresh=1e-3, batch=B, max_trace=32, max_gusto_iters=cap, delta0=delta0, first_solve_cap=cap)
    g.solve_batch(x0, u_init, x_init, z=z)
    info = g.kernel_info
    assert info['family'] == 'lean' and 1 <= info['handed_over'] <= B, info     # rollouts left the lean kernel for the full QP
    assert (g.status == 0).all(), g.status
    calls = []
    real = ripm.solve
    monkeypatch.setattr(ripm, 'solve', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    model = dict(w['tab'], w_q=1.0, w_v=0.0)
    ref = ogusto.solve(model, w['Ad'], w['Bd'], w['dd'], w['H'], w['N'], w['dt'], w['Qz'], w['R'], x0[0], u_init[0], x_init[0], z=z[0],
                       U=(w['UA'], w['Ub']), X=(w['XA'], w['Xb']), x_char=xc, f_char=fc, convg_thresh=1e-3, qp_solver='condensed_ipm',
                       max_gusto_iters=cap, delta0=delta0)
    assert len(calls) >= 5, len(calls)
    compare(g, 0, ref, 'binding trust region')
    assert all(t[1] <= delta0 for t in ref[3])                                  # the trust region never grew back


@pytest.mark.parametrize('N', [5, 50])
def test_keep_solver_state_across_solves_same_results(N):
    """`GuSTO(keep_solver_state=True)` -- the reference's warm_start=True semantics: its persistent cvxpy problem starts every QP, also the
    first one of the next GuSTO.solve, from the previous solution (sofacontrol/scp/locp.py:181) -- only changes where the interior point of a
    solve's first QP STARTS: a series of different problems solved one after the other must give the independent solves' SCP iteration
    counts, status and trajectories (1e-6: same minimisers at the same gap), and the oracle's."""
    import workloads as wl
    from sofacontrol_amd.scp.gusto import GuSTO
    from sofacontrol_amd.utils import Polyhedron
    w = wl.diamond_c2(N=N, dt=0.05)
    P_ = 6
    gm, xc, fc, x0, u_init, x_init, z = problem(w, P_, 2, 1354)
    res = {}
    for keep in (False, True):
        g = GuSTO(gm, N, 0.05, w['Qz'], w['R'], x0[0], u_init[0], x_init[0], z=z[0], U=Polyhedron(w['UA'], w['Ub']), X=Polyhedron(w['XA'], w['Xb']),
                  x_char=xc, f_char=fc, convg_thresh=1e-3, max_trace=0, max_gusto_iters=5, keep_solver_state=keep)
        out = []
        for rep in range(2):
            for b in range(P_):
                g.solve(x0[b], u_init[b], x_init[b], z=z[b])
                out.append((int(g.iters[0]), int(g.status[0]), g.xopt.copy(), g.uopt.copy()))
        res[keep] = out
    for a, k in zip(res[False], res[True]):
        assert a[0] == k[0] and a[1] == k[1], (a[0], k[0], a[1], k[1])
        assert rel(k[2], a[2]) <= 1e-6 and rel(k[3], a[3]) <= 1e-6, (rel(k[2], a[2]), rel(k[3], a[3]))
    for b in range(2):
        xe, ue, ze, tr = oracle_solve(w, xc, fc, x0[b], u_init[b], x_init[b], z[b], 5)
        k = res[True][P_ + b]
        assert k[0] == len(tr)
        assert rel(k[2], xe) <= 1e-5 and rel(k[3], ue) <= 1e-5, (rel(k[2], xe), rel(k[3], ue))


def test_first_solve_cap_limits_the_constructor_solve():
    """The constructor runs one solve at the reference-default cap of 500 SCP iterations whatever `max_gusto_iters` says (gusto.py:142-147);
    `first_solve_cap` lets a caller who only wants the plan built cap that solve too (bench.py: C5)."""
    import workloads as wl
    from sofacontrol_amd.scp.gusto import GuSTO
    from sofacontrol_amd.utils import Polyhedron
    w = wl.diamond_c2(N=5, dt=0.05)
    gm, xc, fc, x0, u_init, x_init, z = problem(w, 4, 2, 1354)
    kw = dict(z=z, U=Polyhedron(w['UA'], w['Ub']), X=Polyhedron(w['XA'], w['Xb']), x_char=xc, f_char=fc, convg_thresh=1e-3, batch=4, max_trace=0)
    g = GuSTO(gm, 5, 0.05, w['Qz'], w['R'], x0, u_init, x_init, max_gusto_iters=3, first_solve_cap=0, **kw)
    assert (g.iters == 1).all(), g.iters               # cap 0: exactly one QP (the real-time iteration)
    g.solve_batch(x0, u_init, x_init, z=z)
    it_user = g.iters.copy()
    assert (it_user >= 2).any() and (it_user <= 4).all(), it_user
    g2 = GuSTO(gm, 5, 0.05, w['Qz'], w['R'], x0, u_init, x_init, max_gusto_iters=3, **kw)
    assert (g2.iters >= 2).any(), g2.iters             # the default constructor solve is not capped at 0
