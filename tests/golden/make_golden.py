"""Generate golden vectors by IMPORTING THE REFERENCE (build container only).

Usage:  python tests/golden/make_golden.py          (writes tests/golden/*.npz)

The reference python (/root/reference) is imported through the stand-in modules of
_ref_import.py and driven on small seeded synthetic inputs; only inputs/outputs (data) are
stored.  The GPU box and the CI never run this script -- they read the committed .npz files.
"""
import io
import os
import sys
import contextlib

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_import  # noqa: E402

_ref_import.install()

from sofacontrol.mor import pod as rpod  # noqa: E402
from sofacontrol.tpwl import tpwl as rtpwl  # noqa: E402
from sofacontrol.lqr import ilqr as rilqr, lqr as rlqr, traj_tracking_lqr as rtt  # noqa: E402
from sofacontrol.scp import gusto as rgusto  # noqa: E402
from sofacontrol.scp.models.tpwl import TPWLGuSTO  # noqa: E402
from sofacontrol.scp import standalone as rsa  # noqa: E402
from sofacontrol import utils as rutils  # noqa: E402
from sofacontrol.tpwl.tpwl_utils import Target  # noqa: E402
from sofacontrol.measurement_models import linearModel  # noqa: E402

from oracle import tpwl as otpwl, locp as olocp  # noqa: E402


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


def small_rom(n_nodes, r, seed):
    rng = np.random.default_rng(seed)
    n_f = 3 * n_nodes
    U, _ = np.linalg.qr(rng.standard_normal((n_f, r)))
    q_ref = rng.uniform(-108, 107, n_f)
    v_ref = 0.01 * rng.standard_normal(n_f)
    return U, q_ref, v_ref


def ref_tpwl(model, U, q_ref, v_ref, Hf, method='nn', discr='zoh', beta=None):
    data = dict(q=model['q'], v=model['v'], u=model['u'], A_c=model['A_c'], B_c=model['B_c'],
                d_c=model['d_c'], rom_info=dict(type='POD', U=U, q_ref=q_ref, v_ref=v_ref))
    params = dict(tpwl_method=method, dist_weights={'q': model['w_q'], 'v': model['w_v']},
                  beta_weighting=beta)
    return rtpwl.TPWLATV(data=data, params=params, Hf=Hf, discr_method=discr)


def g1_pod(out):
    U, q_ref, v_ref = small_rom(100, 8, 0)
    rng = np.random.default_rng(1)
    rom = rpod.POD(dict(U=U, q_ref=q_ref, v_ref=v_ref))
    Xq = q_ref + 5 * rng.standard_normal((5, 300))
    Xv = rng.standard_normal((5, 300))
    Xx = rutils.qv2x(Xq, Xv)
    M = rng.standard_normal((300, 300))
    Mc = sp.coo_matrix(np.where(np.abs(M) > 1.5, M, 0.0))
    Hm = rng.standard_normal((300, 4))
    res = dict(U=U, q_ref=q_ref, v_ref=v_ref, Xq=Xq, Xv=Xv, M=M, Hm=Hm,
               Mc_dense=Mc.toarray(), V=rom.V, x_ref=rom.x_ref,
               proj_q=np.stack([rom.compute_RO_state(qf=x) for x in Xq]),
               proj_v=np.stack([rom.compute_RO_state(vf=x) for x in Xv]),
               proj_x=np.stack([rom.compute_RO_state(xf=x) for x in Xx]),
               UMU=rom.compute_RO_matrix(M), UM=rom.compute_RO_matrix(M, left=True),
               MU=rom.compute_RO_matrix(M, right=True), UMcU=rom.compute_RO_matrix(Mc),
               UH=rom.compute_RO_matrix(Hm, left=True))
    pr = res['proj_x']
    res['lift_q'] = np.stack([rom.compute_FO_state(q=p[8:]) for p in pr])
    res['lift_v'] = np.stack([rom.compute_FO_state(v=p[:8]) for p in pr])
    res['lift_x'] = np.stack([rom.compute_FO_state(x=p) for p in pr])
    # compute_POD on low-rank + noise snapshots (n_f x n_s)
    L = rng.standard_normal((300, 6)) * np.array([50, 20, 8, 3, 1, 0.3])
    S = L @ rng.standard_normal((6, 40)) + 1e-3 * rng.standard_normal((300, 40))
    for tol in (1e-2, 1e-4, 1e-7):
        U_full, Uk, k, Sig = rpod.compute_POD(S, tol)
        res['pod_k_%g' % tol] = k
    res['pod_S'] = S
    res['pod_Sigma'] = Sig
    res['pod_Ufull_abs'] = np.abs(U_full[:, :8])
    # snapshots helpers
    data = dict(q=[q_ref + i * np.ones(300) for i in range(4)], v=[i * np.ones(300) for i in range(4)])
    data['v+'] = [2.0 * i * np.ones(300) for i in range(4)]
    for t in 'qva':
        res['snap_' + t] = rpod.get_snapshots(data, t)
    # singular values of the shipped Diamond POD model (data fixture) and its truncation
    shipped = rutils.load_data(os.path.join(_ref_import.REF, 'examples/diamond/pod_model.pkl'))
    res['shipped_Sigma'] = shipped['Sigma']
    res['shipped_k'] = shipped['POD_info']['U'].shape[1]
    res['shipped_tol'] = shipped['config']['pod_tolerance']
    np.savez_compressed(os.path.join(out, 'g1_pod.npz'), **res)


def make_problem(r, m, P, n_nodes, seed, q_scale=1.0):
    model = otpwl.synthetic_model(r, m, P, seed=seed)
    model['q'] = model['q'] * q_scale
    U, q_ref, v_ref = small_rom(n_nodes, r, seed + 1)
    tip = linearModel(nodes=[n_nodes // 2], num_nodes=n_nodes)
    Hf = tip.C
    return model, U, q_ref, v_ref, Hf


def g3_tpwl(out):
    r, m, P = 4, 3, 7
    model, U, q_ref, v_ref, Hf = make_problem(r, m, P, 20, 10)
    model['w_v'] = 0.5
    rng = np.random.default_rng(11)
    res = {}
    tp = ref_tpwl(model, U, q_ref, v_ref, Hf)
    X = np.concatenate((0.3 * rng.standard_normal((12, r)), 3 * rng.standard_normal((12, r))), axis=1)
    X[3] = np.concatenate((model['v'][2], model['q'][2]))       # exactly on a point
    res['X'] = X
    res['nearest'] = np.array([tp.calc_nearest_point(x) for x in X])
    tpw = ref_tpwl(model, U, q_ref, v_ref, Hf, method='weighting', beta=3.0)
    res['weights'] = np.stack([tpw.calc_weighting_factors(x) for x in X])
    Aw, Bw, dw = zip(*[tpw.get_jacobians(x) for x in X])
    res['Aw'], res['Bw'], res['dw'] = np.stack(Aw), np.stack(Bw), np.stack(dw)
    res['H'] = np.asarray(tp.H)
    res['z_ref'] = np.asarray(tp.z_ref)
    dt = 0.05
    # weighting mode: discretised blended Jacobians and a step-wise rollout (tpwl.py:244-250, 193-216)
    twz = ref_tpwl(model, U, q_ref, v_ref, Hf, method='weighting', beta=3.0, discr='zoh')
    Adw, Bdw, ddw = zip(*[twz.get_jacobians(x, dt=dt) for x in X[:4]])
    res['Adw'], res['Bdw'], res['ddw'] = np.stack(Adw), np.stack(Bdw), np.stack(ddw)
    uw = np.random.default_rng(12).uniform(0, 800, (6, m))
    xw, zw = twz.rollout(0.1 * X[1], uw, dt)
    res['rollw_u'], res['rollw_x'], res['rollw_z'] = uw, xw, zw
    for meth in ('fe', 'be', 'bil', 'zoh'):
        t2 = ref_tpwl(model, U, q_ref, v_ref, Hf, discr=meth)
        quiet(t2.pre_discretize, dt)
        res['Ad_' + meth] = np.stack(t2.A_d)
        res['Bd_' + meth] = np.stack(t2.B_d)
        res['dd_' + meth] = np.stack(t2.d_d)
    quiet(tp.pre_discretize, dt)
    u = rng.uniform(0, 800, (15, m))
    x0 = 0.1 * rng.standard_normal(2 * r)
    xr, zr = tp.rollout(x0, u, dt)
    res['roll_u'], res['roll_x0'], res['roll_x'], res['roll_z'] = u, x0, xr, zr
    gm = TPWLGuSTO(tp)
    xc, fc = gm.get_characteristic_vals()
    res['x_char'], res['f_char'] = xc, fc
    res['dx_char'] = tp.get_characteristic_dx(dt)
    f, A, B = zip(*[gm.get_continuous_dynamics(x, uu) for x, uu in zip(X, u[:12])])
    res['fc'] = np.stack(f)
    res['U'], res['q_ref'], res['v_ref'] = U, q_ref, v_ref
    res['Hf'] = Hf.toarray()
    np.savez_compressed(os.path.join(out, 'g3_tpwl.npz'), **res)


def g4_riccati(out):
    r, m, P = 5, 4, 9
    model, U, q_ref, v_ref, Hf = make_problem(r, m, P, 30, 20)
    tp = ref_tpwl(model, U, q_ref, v_ref, Hf)
    dt = 0.05
    quiet(tp.pre_discretize, dt)
    rng = np.random.default_rng(21)
    n = 2 * r
    res = {}
    H = np.asarray(tp.H)
    Qz = np.diag([0, 0, 0, 100., 100., 0])
    Q = H.T @ Qz @ H + 1e-3 * np.eye(n)
    R = 1e-3 * np.eye(m)
    A, B = tp.A_d[2], tp.B_d[2]
    (L, Pm), _ = quiet(rlqr.solve_riccati, A, B, Q, R)
    res['sr_L'], res['sr_P'] = L, Pm
    K, Pd = rlqr.dare(A, B, Q, R)
    res['dare_K'], res['dare_P'] = K, Pd
    # TrajTrackingLQR (C1 shape: r=5, horizon 10)
    tgt = Target()
    N = 10
    tgt.t = dt * np.arange(N + 1)
    tgt.u = rng.uniform(0, 500, (N + 1, m))
    tgt.x, _ = tp.rollout(0.05 * rng.standard_normal(n), tgt.u[:-1], dt)
    cost = rutils.QuadraticCost(Q=Q, R=R)
    tt = rtt.TrajTrackingLQR(dt, tp, cost)
    Ktt, Ptt = tt.perform_dlqr_recursion(tgt)
    res['tt_t'], res['tt_u'], res['tt_x'] = tgt.t, tgt.u, tgt.x
    res['tt_K'], res['tt_P'], res['tt_xbar'], res['tt_ubar'] = Ktt, Ptt, tt.x_bar, tt.u_bar
    res['Q'], res['R'], res['H'], res['z_ref'] = Q, R, H, np.asarray(tp.z_ref)
    # iLQR backward pass on a fixed trajectory + full solves
    cost = rutils.QuadraticCost(Q=Qz, R=R, Qf=10 * Qz)
    for tag, N in (('c1', 10), ('n30', 30)):
        il = rilqr.iLQR(dt, tp, cost, N)
        th = np.linspace(0, 2 * np.pi * N / 100., N + 1)
        zt = np.zeros((N + 1, 6))
        zt[:, 3] = -2.0 * np.sin(th)
        zt[:, 4] = 1.0 * np.sin(2 * th)
        zt = zt + np.asarray(tp.z_ref)
        il.set_target(zt)
        x0 = 0.02 * rng.standard_normal(n)
        il.rho, il.drho = 0., 0.
        xp = np.zeros((N + 1, n)); xp[0] = x0
        uw = rng.uniform(0, 100, (N, m))
        (x, u, c, Aj, Bj, dj), _ = quiet(il.forward_pass, xp, uw)
        (K, k, Q_u, Q_uu), _ = quiet(il.dlqr_recursion, x, u, Aj, Bj, dj)
        res[tag + '_z_target'], res[tag + '_x0'], res[tag + '_uw'] = zt, x0, uw
        res[tag + '_fp_x'], res[tag + '_fp_u'], res[tag + '_fp_cost'] = x, u, c
        res[tag + '_K'], res[tag + '_k'], res[tag + '_Qu'], res[tag + '_Quu'] = K, k, Q_u, Q_uu
        res[tag + '_rho_after'] = il.rho
        (xs, us, Ks), log = quiet(il.ilqr_computation, x0, uw)
        res[tag + '_sol_x'], res[tag + '_sol_u'], res[tag + '_sol_K'] = xs, us, Ks
        res[tag + '_iters'] = log.count('Iteration')
        (xs0, us0, Ks0), log = quiet(il.ilqr_computation, x0)
        res[tag + '_sol0_x'], res[tag + '_sol0_u'], res[tag + '_sol0_K'] = xs0, us0, Ks0
        res[tag + '_iters0'] = log.count('Iteration')
    res['Qz'], res['Qf'] = Qz, 10 * Qz
    np.savez_compressed(os.path.join(out, 'g4_riccati.npz'), **res)


def g20_ilqr_switches(out):
    """The four switches of lqr/config.py:6-9, 31, one at a time and all together, on the g4 problem (r = 5, n_u = 4, N = 30):
    what the imported reference iLQR returns with include_input_var_constraint / do_linesearch / regularize /
    state_regularization = False."""
    r, m, P = 5, 4, 9
    model, U, q_ref, v_ref, Hf = make_problem(r, m, P, 30, 20)
    tp = ref_tpwl(model, U, q_ref, v_ref, Hf)
    dt, N = 0.05, 30
    quiet(tp.pre_discretize, dt)
    rng = np.random.default_rng(77)
    n = 2 * r
    Qz = np.diag([0, 0, 0, 100., 100., 0])
    R = 1e-3 * np.eye(m)
    cost = rutils.QuadraticCost(Q=Qz, R=R, Qf=10 * Qz)
    th = np.linspace(0, 2 * np.pi * N / 100., N + 1)
    zt = np.zeros((N + 1, 6))
    zt[:, 3] = -2.0 * np.sin(th)
    zt[:, 4] = 1.0 * np.sin(2 * th)
    zt = zt + np.asarray(tp.z_ref)
    x0 = 0.02 * rng.standard_normal(n)
    uw = rng.uniform(0, 100, (N, m))
    u_last = rng.uniform(0, 100, m)
    res = dict(z_target=zt, x0=x0, uw=uw, u_last=u_last, Qz=Qz, R=R, Qf=10 * Qz, dt=dt, N=N)
    flags = ('include_input_var_constraint', 'do_linesearch', 'regularize', 'state_regularization')
    cases = {f: {f: False} for f in flags}
    cases['all_off'] = {f: False for f in flags}
    cases['reference'] = {}
    for tag, off in cases.items():
        for warm in (True, False):
            il = rilqr.iLQR(dt, tp, cost, N)
            for f, v in off.items():
                setattr(il.params, f, v)
            il.set_target(zt)
            il.set_u_last(u_last)
            (xs, us, Ks), log = quiet(il.ilqr_computation, x0, uw if warm else None)
            key = tag + ('_warm' if warm else '_cold')
            res[key + '_x'], res[key + '_u'], res[key + '_K'] = xs, us, Ks
            res[key + '_iters'] = log.count('Iteration')
    np.savez_compressed(os.path.join(out, 'g20_ilqr_switches.npz'), **res)


class InjectedLOCP:
    """Stand-in for sofacontrol.scp.locp.LOCP (cvxpy is absent): same update/solve/get_solution
    protocol (locp.py:98,175,192), QP data from oracle.locp.build_qp, solved exactly."""
    log = []

    def __init__(self, N, H, Qz, R, Qzf=None, U=None, X=None, Xf=None, dU=None, verbose=False,
                 warm_start=True, x_char=None, **kwargs):
        self.N, self.H, self.Qz, self.R, self.Qzf = N, H, Qz, R, Qzf
        cv = lambda p: None if p is None else (np.asarray(p.A), np.asarray(p.b))
        self.U, self.X, self.Xf, self.dU = cv(U), cv(X), cv(Xf), cv(dU)
        self.xs = np.ones(H.shape[1]) if x_char is None else 1. / np.abs(x_char)

    def update(self, Ad, Bd, dd, x0, xk, delta, omega, z=None, zf=None, u=None, full=True, **kw):
        if full:
            self.args = (Ad, Bd, dd, np.asarray(x0), np.asarray(xk))
            self.z, self.zf, self.u = z, zf, u
        self.delta, self.omega = delta, omega

    def solve(self):
        Ad, Bd, dd, x0, xk = self.args
        qp = olocp.build_qp(self.N, self.H, self.Qz, self.R, Ad, Bd, dd, x0, xk, self.delta,
                            self.omega, z=self.z, u_des=self.u, Qzf=self.Qzf, zf=self.zf, U=self.U,
                            X=self.X, Xf=self.Xf, dU=self.dU, x_scale=self.xs)
        w, _, _ = olocp.solve_exact(qp)
        self.qp, self.w = qp, w
        J = olocp.objective(qp, w)
        InjectedLOCP.log.append((J, self.delta, self.omega))

        class S:
            solve_time = 0.0
        return J, True, S

    def get_solution(self):
        return olocp.split(self.qp, self.w)


def g6_gusto(out):
    rgusto.LOCP = InjectedLOCP
    res = {}
    r, m, P = 4, 3, 7
    model, U, q_ref, v_ref, Hf = make_problem(r, m, P, 20, 30, q_scale=0.05)
    tp = ref_tpwl(model, U, q_ref, v_ref, Hf)
    dt, N = 0.05, 12
    gm = TPWLGuSTO(tp)
    quiet(gm.pre_discretize, dt)
    n = 2 * r
    H = np.asarray(tp.H)
    Qz = np.diag([0, 0, 0, 100., 100., 0])
    R = 1e-5 * np.eye(m)
    rng = np.random.default_rng(31)
    x_char, f_char = gm.get_characteristic_vals()
    # steady-state reachable amplitude of the tip for this model: used to scale the target
    T = 3.0
    t = np.linspace(0, T, 300)
    th = np.linspace(0, 2 * np.pi, 300)
    zt = np.zeros((300, 6))
    amp = 0.15
    zt[:, 3] = -amp * np.sin(th)
    zt[:, 4] = 0.5 * amp * np.sin(2 * th)
    Ubox = rutils.HyperRectangle([800.] * m, [0.] * m)
    Hz = np.zeros((2, 6)); Hz[0, 3] = 1; Hz[1, 4] = 1
    Hx = Hz @ H
    Xp = rutils.Polyhedron(A=np.vstack([-Hx, Hx]), b=np.array([0.02, 0.02, 0.04, 0.03]))
    x0 = np.zeros(n)
    cases = dict(box=dict(U=Ubox), boxX=dict(U=Ubox, X=Xp), free=dict())
    for tag, cons in cases.items():
        InjectedLOCP.log = []
        node, _ = quiet(rsa.GuSTOSolverNode, gm, N, dt, Qz, R, x0, t=t, z=zt, verbose=0,
                        warm_start=True, convg_thresh=1e-3, jit=False, **cons)
        xopt, uopt, zopt, topt = node.get_solution()
        res[tag + '_xopt'], res[tag + '_uopt'], res[tag + '_zopt'] = xopt, uopt, zopt
        res[tag + '_trace'] = np.array(InjectedLOCP.log)
        # one warm-started receding-horizon re-solve, restating the shift of ros.py:109-114
        t0 = 2 * dt
        x0b = xopt[2] + 1e-3 * rng.standard_normal(n)
        zb, zfb, ub = node.get_target(t0)
        idx0 = np.argwhere(topt >= t0)[0, 0]
        u_init = uopt[-1, :].reshape(1, -1).repeat(N, axis=0)
        u_init[0:N - idx0] = uopt[idx0:, :]
        x_init = xopt[-1, :].reshape(1, -1).repeat(N + 1, axis=0)
        x_init[0:N + 1 - idx0] = xopt[idx0:, :]
        InjectedLOCP.log = []
        node.gusto.max_gusto_iters = 500
        quiet(node.gusto.solve, x0b, u_init, x_init, z=zb, zf=zfb, u=ub)
        x2, u2, z2, _ = node.gusto.get_solution()
        res[tag + '_x0b'], res[tag + '_zb'] = x0b, zb
        res[tag + '_uinit'], res[tag + '_xinit'] = u_init, x_init
        res[tag + '_xopt2'], res[tag + '_uopt2'], res[tag + '_zopt2'] = x2, u2, z2
        res[tag + '_trace2'] = np.array(InjectedLOCP.log)
    # helper functions on fixed data
    g = node.gusto
    xa = xopt + 0.3 * x_char * rng.standard_normal(xopt.shape)
    ua = uopt + 30.0 * rng.standard_normal(uopt.shape)
    g.x_k, g.u_k = xopt.copy(), uopt.copy()
    res['h_x'], res['h_u'], res['h_xk'], res['h_uk'] = xa, ua, xopt, uopt
    res['h_tr'] = np.array([g.is_in_trust_region(xa, d)[0] for d in (1e-3, 1e-1, 10.)])
    res['h_conv'] = g.is_converged(xa, ua)[0]
    res['h_rho'] = g.compute_accuracy(xa, ua, 3.7)
    g.X = Xp
    res['h_viol'] = g.state_constraints_violated(10 * xa)[0]
    res['x_char'], res['f_char'] = x_char, f_char
    res['t'], res['zt'] = t, zt
    res['Xp_A'], res['Xp_b'] = Xp.A, Xp.b
    res['U_A'], res['U_b'] = Ubox.A, Ubox.b
    res['H'], res['Qz'], res['R'] = H, Qz, R
    res['get_target_z'], _, _ = node.get_target(0.37)
    np.savez_compressed(os.path.join(out, 'g6_gusto.npz'), **res)


    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))


def meas_selector(nodes, num_nodes):
    """Position rows of the listed nodes out of x = [v; q] (same structure as measurement_models.linearModel)."""
    n_f = 3 * num_nodes
    Cf = sp.lil_matrix((3 * len(nodes), 2 * n_f))
    for i, nd in enumerate(nodes):
        for a in range(3):
            Cf[3 * i + a, n_f + 3 * nd + a] = 1.0
    return Cf.tocsr()


def g9_ekf(out):
    """DiscreteEKFObserver (tpwl/observer.py:33-126) over the synthetic TPWL model."""
    from sofacontrol.tpwl.observer import DiscreteEKFObserver
    r, m, P, n_nodes = 4, 3, 7, 20
    model, U, q_ref, v_ref, Hf = make_problem(r, m, P, n_nodes, 50, q_scale=0.3)
    Cf = meas_selector([3, 9], n_nodes)
    tp = rtpwl.TPWLATV(data=dict(q=model['q'], v=model['v'], u=model['u'], A_c=model['A_c'], B_c=model['B_c'],
                                 d_c=model['d_c'], rom_info=dict(type='POD', U=U, q_ref=q_ref, v_ref=v_ref)),
                       params=dict(tpwl_method='nn', dist_weights={'q': 1.0, 'v': 0.0}, beta_weighting=None),
                       Cf=Cf, Hf=Hf, discr_method='zoh')
    dt = 0.02
    quiet(tp.pre_discretize, dt)
    n = 2 * r
    rng = np.random.default_rng(51)
    G = rng.standard_normal((n, n))
    W = 10 * np.eye(n) + 0.1 * (G + G.T)
    Vn = 0.1 * np.eye(6) + 0.01 * np.diag(rng.uniform(0, 1, 6))
    Sigma0 = 0.5 * np.eye(n)
    ekf = DiscreteEKFObserver(tp, Sigma0=Sigma0.copy(), W=W, V=Vn)
    res = dict(W=W, V=Vn, Sigma0=Sigma0, Cf=Cf.toarray(), x_init=ekf.x.copy(), z_init=np.asarray(ekf.z).copy())
    xs, Ss, zs, us, ys = [], [], [], [], []
    xt = 0.05 * rng.standard_normal(n)
    for k in range(8):
        u = rng.uniform(0, 500, m)
        A, B, d = tp.get_jacobians(xt, dt)
        xt = A @ xt + B @ u + d
        y = np.asarray(tp.C @ xt).ravel() + tp.y_ref + 0.01 * rng.standard_normal(6)
        ekf.update(u, y, dt)
        us.append(u); ys.append(y); xs.append(ekf.x.copy()); Ss.append(ekf.Sigma.copy()); zs.append(np.asarray(ekf.z).copy())
    res.update(u=np.stack(us), y=np.stack(ys), x=np.stack(xs), Sigma=np.stack(Ss), z=np.stack(zs))
    ekf.predict_state(us[0], dt)
    res['x_pred'], res['Sigma_pred'] = ekf.x.copy(), ekf.Sigma.copy()
    ekf.update_state(ys[1])
    res['x_upd'], res['Sigma_upd'] = ekf.x.copy(), ekf.Sigma.copy()
    xf0 = np.concatenate((v_ref, q_ref)) + 0.01 * rng.standard_normal(6 * n_nodes)
    ekf.initialize(xf0)
    res['xf0'], res['x_reinit'] = xf0, ekf.x.copy()
    np.savez_compressed(os.path.join(out, 'g9_ekf.npz'), **res)


def _mat(v):
    a = np.empty((1, 1), dtype=object)
    a[0, 0] = np.asarray(v)
    return a


def ref_ssm(model, discrete=False, discr='fe'):
    """The reference's SSMDynamics on an oracle.ssm model dict (.mat-style nested arrays, ssm.py:31-52)."""
    from sofacontrol.SSM import ssm as rssm
    sc = lambda v: _mat(np.array([[v]]))
    n, m = model['n'], model['m']
    ro, so = int(model['Er'].sum(axis=1).max()), int(model['Es'].sum(axis=1).max())
    params = dict(state_dim=sc(n), input_dim=sc(m), output_dim=sc(n), SSM_order=sc(so), ROM_order=sc(ro))
    mdl = dict(Ts=sc(0.01), w_coeff=_mat(model['W']), v_coeff=_mat(model['V']), r_coeff=_mat(model['R']),
               B=_mat(model['B']), rd_coeff=_mat(model['Rd']), Bd=_mat(model['Bd']))
    return rssm.SSMDynamics(model['z_ref'].copy(), discrete=discrete, discr_method=discr, model=mdl, params=params)


def g10_ssm(out):
    """SSM polynomial model (SSM/ssm.py) -- reference run with numpy for jax.numpy and complex-step
    differentiation for jax.jacobian (see _ref_import.py)."""
    from oracle import ssm as ossm
    from sofacontrol.scp.models.ssm import SSMGuSTO
    res = {}
    for tag, (n, m, ro, so) in dict(a=(4, 2, 3, 2), b=(6, 4, 3, 3)).items():
        model = ossm.synthetic(n, m, ro, so, seed=60 + n)
        rng = np.random.default_rng(61 + n)
        X = 0.4 * rng.standard_normal((5, n))
        Uu = rng.standard_normal((5, m))
        dt = 0.01
        s = ref_ssm(model)
        res[tag + '_X'], res[tag + '_U'] = X, Uu
        res[tag + '_phi_rom'] = np.stack([np.asarray(s.rom_phi(*x)) for x in X])
        res[tag + '_phi_ssm'] = np.stack([np.asarray(s.ssm_phi(*x)) for x in X])
        Ac, Bc, dc = zip(*[s.get_continuous_jacobians(x, u) for x, u in zip(X, Uu)])
        res[tag + '_Ac'], res[tag + '_Bc'], res[tag + '_dc'] = np.stack(Ac), np.stack(Bc), np.stack(dc)
        res[tag + '_f'] = np.stack([np.asarray(s.reduced_dynamics(x, u)) for x, u in zip(X, Uu)])
        Hs, cs = zip(*[s.get_observer_jacobians(x) for x in X])
        res[tag + '_Hobs'], res[tag + '_cobs'] = np.stack(Hs), np.stack(cs)
        res[tag + '_zobs'] = np.stack([s.update_observer_state(x) for x in X])
        res[tag + '_zf'] = s.x_to_zfyf(X)
        res[tag + '_xred'] = np.stack([np.asarray(s.compute_RO_state(z)) for z in res[tag + '_zf']])
        uu = rng.standard_normal((6, m))
        res[tag + '_roll_u'] = uu
        for meth in ('fe', 'be', 'bil'):
            sm = ref_ssm(model, discr=meth)
            A, B, d = zip(*[sm.get_jacobians(x, u, dt) for x, u in zip(X, Uu)])
            res[tag + '_Ad_' + meth], res[tag + '_Bd_' + meth], res[tag + '_dd_' + meth] = \
                np.stack(A), np.stack(B), np.stack(d)
            xr, zr = sm.rollout(X[0], uu, dt)
            res[tag + '_roll_x_' + meth], res[tag + '_roll_z_' + meth] = xr, zr
        sd = ref_ssm(model, discrete=True)
        A, B, d = zip(*[sd.get_jacobians(x, u, dt) for x, u in zip(X, Uu)])
        res[tag + '_Ad_map'], res[tag + '_Bd_map'], res[tag + '_dd_map'] = np.stack(A), np.stack(B), np.stack(d)
        xr, zr = sd.rollout(X[0], uu, dt)
        res[tag + '_roll_x_map'], res[tag + '_roll_z_map'] = xr, zr
        gm = SSMGuSTO(s)
        res[tag + '_fc'] = np.stack([gm.get_continuous_dynamics(x, u)[0] for x, u in zip(X, Uu)])
        try:
            ref_ssm(model, discr='zoh').get_jacobians(X[0], Uu[0], dt)
            res[tag + '_zoh_raises'] = np.array(0)
        except RuntimeError:
            res[tag + '_zoh_raises'] = np.array(1)
    np.savez_compressed(os.path.join(out, 'g10_ssm.npz'), **res)


def g11_ilqr_ssm(out):
    """iLQR (lqr/ilqr.py) over the reference's SSMDynamics (C3 shape family: SSM model, long horizon), with the
    model's default H = 0 (ssm.py:69-70) and with H set to the linear part of the observer map."""
    from oracle import ssm as ossm
    res = {}
    n, m = 6, 4
    model = ossm.synthetic(n, m, 3, 3, seed=90)
    rng = np.random.default_rng(91)
    dt = 0.01
    for tag, (meth, N, useH) in dict(h0=('be', 20, False), hw=('be', 40, True), fe=('fe', 25, True)).items():
        s = ref_ssm(model, discr=meth)
        if useH:
            s.H = model['W'][:, :n].copy()
        Qz = np.diag([100., 100., 10., 0., 0., 1.])
        cost = rutils.QuadraticCost(Q=Qz, R=0.05 * np.eye(m), Qf=5 * Qz)
        il = rilqr.iLQR(dt, s, cost, N)
        th = np.linspace(0, 2 * np.pi * N / 60., N + 1)
        zt = np.zeros((N + 1, n))
        zt[:, 0] = 0.3 * np.sin(th)
        zt[:, 1] = 0.2 * (1 - np.cos(th))
        zt = zt + model['z_ref']
        il.set_target(zt)
        x0 = 0.05 * rng.standard_normal(n)
        uw = 0.1 * rng.standard_normal((N, m))
        (xs, us, Ks), log = quiet(il.ilqr_computation, x0, uw)
        res[tag + '_z_target'], res[tag + '_x0'], res[tag + '_uw'] = zt, x0, uw
        res[tag + '_x'], res[tag + '_u'], res[tag + '_K'] = xs, us, Ks
        res[tag + '_iters'] = np.array(log.count('Iteration'))
        res[tag + '_Qz'] = Qz
    np.savez_compressed(os.path.join(out, 'g11_ilqr_ssm.npz'), **res)


def g12_assembly(out):
    """TPWLSnapshotData.add_point / add_continuous_TPWL / add_discrete_TPWL / evaluate_point_dist
    (tpwl/tpwl_utils.py:84-117, 170-196, 263-290) on synthetic full-order points."""
    from sofacontrol.tpwl.tpwl_utils import TPWLSnapshotData
    from types import SimpleNamespace
    n_nodes, r, m = 30, 5, 3
    U, q_ref, v_ref = small_rom(n_nodes, r, 120)
    n_f = 3 * n_nodes
    rom = rpod.POD(dict(U=U, q_ref=q_ref, v_ref=v_ref))
    cfg = SimpleNamespace(eval_type='distance', save_continuous_TPWL=True, save_discrete_TPWL=True,
                          TPWL_weighting_factors={'q': 1.0, 'v': 0.2}, TPWL_separate_calculation=False,
                          TPWL_threshold=1.0, TPWL_type='ATV', discr_type='zoh')
    data = TPWLSnapshotData(rom, cfg)
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import assembly_points
    res = {}
    pts = []
    for d in assembly_points(n_f, m, q_ref, 121):
        p = rutils.Point()
        for k, v in d.items():
            setattr(p, k, v)
        pts.append(p)
    res['eval0'] = np.array(data.evaluate_point(pts[0], None))
    quiet(data.add_point, pts[0])
    res['eval1'] = np.array(data.evaluate_point(pts[1], pts[0]))
    quiet(data.add_point, pts[1])
    near = rutils.Point(); near.q = pts[1].q + 1e-3; near.v = pts[1].v
    res['eval_near'] = np.array(data.evaluate_point(near, pts[1]))
    quiet(data.add_point, pts[2])
    for k in ('q', 'v', 'K', 'D', 'M', 'S', 'H', 'b', 'f', 'q+', 'v+', 'A_c', 'B_c', 'd_c', 'A_d', 'B_d', 'd_d'):
        res['out_' + k] = np.asarray(data.dict[k])
    np.savez_compressed(os.path.join(out, 'g12_assembly.npz'), **res)


def g13_controllers2(out):
    """TrajTracking and StateDLQR controllers (tpwl/controllers.py:349-437) driven by a scripted
    (sim_time, y, x, u_prev) sequence."""
    import sofacontrol.tpwl.controllers as ctl
    from sofacontrol.tpwl.tpwl_utils import DynamicsTarget
    r, m, P = 4, 3, 7
    model, U, q_ref, v_ref, Hf = make_problem(r, m, P, 20, 40, q_scale=0.05)
    tp = ref_tpwl(model, U, q_ref, v_ref, Hf)
    n = 2 * r
    rng = np.random.default_rng(130)
    dt = 0.02
    res = {}
    H = np.asarray(tp.H)
    cost = rutils.QuadraticCost()
    cost.Q = H.T @ np.diag([0, 0, 0, 100., 100., 0]) @ H + 1e-2 * np.eye(n)
    cost.R = 1e-3 * np.eye(m)
    tgt = Target()
    Nt = 12
    tgt.t = dt * np.arange(Nt + 1)
    tgt.u = rng.uniform(0, 300, (Nt + 1, m))
    tgt.x, _ = tp.rollout(0.02 * rng.standard_normal(n), tgt.u[:-1], dt)
    (c, _) = quiet(ctl.TrajTracking, tp, cost, tgt, dt=dt, delay=0.02)
    c.set_sim_timestep(dt)
    V = np.kron(np.eye(2), U)
    x_ref = np.concatenate((v_ref, q_ref))
    xs = [V @ (tgt.x[min(k, Nt)] + 1e-3 * rng.standard_normal(n)) + x_ref for k in range(13)]
    us = [quiet(c.evaluate, k * dt, None, xs[k], np.zeros(m))[0] for k in range(13)]
    res['tt_t'], res['tt_u_target'], res['tt_x_target'] = tgt.t, tgt.u, tgt.x
    res['tt_x_full'], res['tt_u'] = np.stack(xs), np.stack(us)
    res['tt_K'] = np.asarray(c.K)
    # StateDLQR around the stored point 2
    dtg = DynamicsTarget()
    dtg.A, dtg.B = model['A_c'][2], model['B_c'][2]
    dtg.x = np.concatenate((model['v'][2], model['q'][2]))
    dtg.u = model['u'][2]
    (c2, _) = quiet(ctl.StateDLQR, tp, cost, dtg, dt=dt, delay=0.0)
    c2.set_sim_timestep(dt)
    xs2 = [V @ (dtg.x + 1e-3 * rng.standard_normal(n)) + x_ref for k in range(4)]
    res['dl_x_full'] = np.stack(xs2)
    res['dl_u'] = np.stack([quiet(c2.evaluate, k * dt, None, xs2[k], np.zeros(m))[0] for k in range(4)])
    res['dl_K'] = np.asarray(c2.K)
    res['Q'], res['R'] = cost.Q, cost.R
    np.savez_compressed(os.path.join(out, 'g13_controllers2.npz'), **res)


class FakeGuSTOClient:
    """Deterministic stand-in for GuSTOClientNode (needs ROS): returns a smooth analytic 'solution'."""
    N, dt_g = 8, 0.05

    def __init__(self):
        self.done = False

    def send_request(self, t0, x0, wait=True):
        self.t0, self.x0 = float(t0), np.asarray(x0, dtype=float).copy()
        self.done = True

    def force_spin(self):
        pass

    def check_if_done(self):
        return self.done

    def force_wait(self):
        pass

    def get_solution(self, n_x, n_u):
        t = self.t0 + self.dt_g * np.arange(self.N + 1)
        x = np.stack([self.x0 * np.cos(3 * (tt - self.t0)) + 0.01 * np.sin(tt + np.arange(n_x)) for tt in t])
        u = np.stack([50.0 + 40.0 * np.sin(2 * tt + np.arange(n_u)) for tt in t[:-1]])
        return t, u, x, 0.0123


def g8_controllers(out):
    import sofacontrol.tpwl.controllers as ctl
    ctl.GuSTOClientNode = FakeGuSTOClient
    r, m, P = 4, 3, 7
    model, U, q_ref, v_ref, Hf = make_problem(r, m, P, 20, 40, q_scale=0.05)
    tp = ref_tpwl(model, U, q_ref, v_ref, Hf)
    n = 2 * r
    H = np.asarray(tp.H)
    cost = rutils.QuadraticCost()
    cost.Q = H.T @ np.diag([0, 0, 0, 100., 100., 0]) @ H + 1e-2 * np.eye(n)
    cost.R = 1e-3 * np.eye(m)
    dt = 0.01
    (c, _) = quiet(ctl.scp, tp, cost, dt, N_replan=3, delay=0.02)
    c.set_sim_timestep(dt)
    rng = np.random.default_rng(41)
    n_f = 60
    x_ref = rutils.qv2x(q_ref, v_ref)
    us, xs = [], []
    steps = 14

    def run():
        for k in range(steps):
            xf = x_ref + 0.2 * rng.standard_normal(2 * n_f)
            xs.append(xf)
            us.append(c.evaluate(k * dt, None, xf, np.zeros(m)))
    quiet(run)
    info = c.save_controller_info()
    res = dict(x_full=np.stack(xs), u=np.stack(us), t_opt=info['t_opt'], u_opt=info['u_opt'], z_opt=info['z_opt'],
               K=np.stack(c.K), Q=cost.Q, R=cost.R, rollout_time=info['rollout_time'], n_solves=len(info['solve_times']))
    # ilqr controller: trajectory-tracking target
    tgt = Target()
    tgt.t = np.linspace(0, 0.2, 21)
    tgt.z = np.zeros((21, 6)); tgt.z[:, 3] = -1.0 * np.sin(10 * tgt.t); tgt.z[:, 4] = 0.5 * np.sin(20 * tgt.t)
    tgt.z = tgt.z + np.asarray(tp.z_ref)
    tgt.Hf = Hf
    cost2 = rutils.QuadraticCost(Q=np.diag([0, 0, 0, 100., 100., 0]), R=1e-3 * np.eye(m), Qf=np.diag([0, 0, 0, 100., 100., 0]))
    quiet(tp.pre_discretize, 0.02)
    (ci, _) = quiet(ctl.ilqr, tp, cost2, tgt, dt=0.02, delay=0.0)
    ci.set_sim_timestep(0.01)
    us2 = []

    def run2():
        for k in range(8):
            us2.append(ci.evaluate(k * 0.01, None, xs[k], np.zeros(m)))
    quiet(run2)
    res['il_u'] = np.stack(us2)
    res['il_t'], res['il_z'] = tgt.t, tgt.z
    res['il_xbar'], res['il_ubar'] = ci.x_bar, ci.u_bar
    np.savez_compressed(os.path.join(out, 'g8_controllers.npz'), **res)


def g15_ssm_controllers(out):
    """SSM closed-loop controller (SSM/controllers.py:16-252: TemplateController.evaluate time gating, SSMObserver
    302-310, scp policy stitching 140-252) driven by a scripted measurement sequence, the solver client replaced by the
    deterministic stand-in above (the real one needs ROS).  Y = None: the re-projection of measurements
    (controllers.py:96-97) goes through OSQP, absent here -- the product's projection is tested against the exact
    projection instead (tests/test_edge_cases_gpu.py)."""
    import sofacontrol.SSM.controllers as sctl
    from oracle import ssm as ossm
    sctl.GuSTOClientNode = FakeGuSTOClient
    n, m = 6, 4
    model = ossm.synthetic(n, m, 3, 3, seed=150)
    s = ref_ssm(model)
    rng = np.random.default_rng(151)
    dt = 0.01
    res = {}
    for tag, (N_replan, delay, steps) in dict(a=(3, 0.02, 14), b=(1, 0.0, 6)).items():
        (c, _) = quiet(sctl.scp, s, None, dt, N_replan=N_replan, delay=delay)
        c.set_sim_timestep(dt)
        ys, us, xs = [], [], []

        def run():
            for k in range(steps):
                y = rutils.vq2qv(model['z_ref'] + 0.05 * rng.standard_normal(n))      # swaps the halves: its own inverse
                ys.append(y)
                us.append(c.evaluate(k * dt, y, None, np.zeros(m)))
                xs.append(np.asarray(c.observer.x).copy())
        quiet(run)
        info = c.save_controller_info()
        res[tag + '_y'], res[tag + '_u'], res[tag + '_x_obs'] = np.stack(ys), np.stack(us), np.stack(xs)
        res[tag + '_z_obs'] = np.asarray(c.observer.z)
        res[tag + '_t_opt'], res[tag + '_u_opt'], res[tag + '_z_opt'] = info['t_opt'], info['u_opt'], info['z_opt']
        res[tag + '_rollout_time'] = info['rollout_time']
        res[tag + '_n_solves'] = len(info['solve_times'])
        res[tag + '_z_rollout0'] = np.asarray(info['z_rollout'][0])
        res[tag + '_t_rollout0'] = np.asarray(info['t_rollout'][0])
        res[tag + '_params'] = np.array([N_replan, delay, steps, dt])
    np.savez_compressed(os.path.join(out, 'g15_ssm_controllers.npz'), **res)


def g16_dubins(out):
    """The reference's own runnable GuSTO demonstration (sofacontrol/scp/example.py:1-35): DubinsCar, N = 50, dt = 0.1,
    terminal cost only, input-rate polytope dU, x_char, warm_start=False -- run with the imported reference GuSTO class and
    model, the QP solved by the exact oracle in place of cvxpy; plus a second case with the U and X boxes that
    example.py defines but does not pass (its terminal box Xf is infeasible for the first linearisation: y is not
    reachable from a straight, standing start)."""
    from sofacontrol.scp.models.dubins_car import DubinsCar
    rgusto.LOCP = InjectedLOCP
    model = DubinsCar()
    model.nonlinear_observer = False     # gusto.py:133 reads it; the reference's DubinsCar never sets it (example.py fails there)
    N, dt = 50, 0.1
    U = rutils.HyperRectangle(np.array([1., 1.]), np.array([0., -1.]))
    xmax = np.array([6., 6., np.pi])
    X = rutils.HyperRectangle(xmax, -xmax)
    x_target = np.array([5., 5., 0.])
    Xf = rutils.HyperRectangle(x_target + 2, x_target - 2)
    dU = rutils.HyperRectangle(np.array([0.1, 0.1]), np.array([-0.1, -0.1]))
    Qz = np.zeros((3, 3)); R = np.eye(2); Qzf = 100 * np.eye(3)
    zf_des = np.array([4., 5., 0.])
    x0 = np.zeros(3)
    u_init = np.zeros((N, 2))
    x_init = model.rollout(x0, u_init, dt)
    x_char = np.array([1., 1., np.pi])
    res = dict(N=N, dt=dt, Qz=Qz, R=R, Qzf=Qzf, zf=zf_des, x0=x0, u_init=u_init, x_init=x_init, x_char=x_char)
    for tag, cons in dict(example=dict(U=None, dU=dU), boxes=dict(U=U, X=X, dU=dU)).items():
        InjectedLOCP.log = []
        g, _ = quiet(rgusto.GuSTO, model, N, dt, Qz, R, x0, u_init, x_init, u=u_init, zf=zf_des, Qzf=Qzf,
                     verbose=0, visual=[], warm_start=False, x_char=x_char, jit=False, **cons)
        x, u, z, _ = g.get_solution()
        res[tag + '_x'], res[tag + '_u'], res[tag + '_z'] = x, u, z
        res[tag + '_trace'] = np.array(InjectedLOCP.log)
    for nm, poly in dict(U=U, X=X, Xf=Xf, dU=dU).items():
        res[nm + '_A'], res[nm + '_b'] = poly.A, poly.b
    np.savez_compressed(os.path.join(out, 'g16_dubins.npz'), **res)


def g17_ssm_hardware(out):
    """The reference's own SSM module test (examples/hardware/diamond_SSM.py:21-80, `module_test_continuous` and its
    continuous twin): the SHIPPED Diamond SSM model (examples/hardware/SSMmodels/SSM_model.mat: n_x = 6, n_u = 4, cubic,
    83 monomials) rolled out open loop on the recorded inputs (checkModel/u_big.csv, 1002 x 4) and compared with the
    recorded tip trajectory (checkModel/z_big.csv, [v; q] rows).  The fixture holds the reference's data files (model
    coefficients, inputs, measured outputs) and the trajectories / mean squared errors the imported reference computes
    (numpy standing in for jax.numpy)."""
    from scipy.io import loadmat
    from scipy.interpolate import interp1d
    from sofacontrol.SSM import ssm as rssm
    from _ref_import import REF
    base = os.path.join(REF, 'examples', 'hardware')
    data = loadmat(os.path.join(base, 'SSMmodels', 'SSM_model.mat'))['py_data'][0, 0]
    raw_model, raw_params = data['model'], data['params']
    u_true = np.genfromtxt(os.path.join(base, 'checkModel', 'u_big.csv'), delimiter=',')
    z_true = np.genfromtxt(os.path.join(base, 'checkModel', 'z_big.csv'), delimiter=',')
    zq_true, zv_true = rutils.x2qv(z_true)
    # equilibrium of the tip exactly as the module test takes it (diamond_SSM.py:29-36): rest_qv.pkl -> x_eq -> tip node 1354
    rest = rutils.load_data(os.path.join(base, 'rest_qv.pkl'))['rest']
    x_eq = rutils.qv2x(q=np.asarray(rest[0]), v=np.asarray(rest[1]))
    z_eq = linearModel([1354], 1628).evaluate(x_eq, qv=True)
    dt, T = 0.01, 10.01
    N = int(T / dt)
    t_original = np.linspace(0, T, int(T / 0.01) + 1)
    t_interp = np.linspace(0, T, N + 1)
    u_interp = interp1d(t_original, u_true, axis=0)(t_interp)
    z_true_qv = interp1d(t_original, np.hstack((zq_true, zv_true)), axis=0)(t_interp)
    res = dict(u_big=u_true, z_big=z_true, z_eq=z_eq, dt=dt, u_interp=u_interp, z_true_qv=z_true_qv)
    for k in raw_model.dtype.names:
        res['model_' + k] = np.asarray(raw_model[k][0, 0])
    for k in raw_params.dtype.names:
        res['params_' + k] = np.asarray(raw_params[k][0, 0])
    n = int(raw_model['B'][0, 0].shape[0])
    for tag, kw in dict(discrete=dict(discrete=True, discr_method='be'), be=dict(discrete=False, discr_method='be'),
                        fe=dict(discrete=False, discr_method='fe')).items():
        model = rssm.SSMDynamics(z_eq, model=raw_model, params=raw_params, **kw)
        p_traj, z_traj = model.rollout(np.zeros(n), u_interp, dt)
        err = z_true_qv - np.asarray(z_traj)[:-1]
        res[tag + '_p'], res[tag + '_z'] = np.asarray(p_traj), np.asarray(z_traj)
        res[tag + '_mse'] = np.linalg.norm(np.linalg.norm(err, axis=1)) ** 2 / err.shape[0]
    np.savez_compressed(os.path.join(out, 'g17_ssm_hardware.npz'), **res)


def g18_pod_shipped(out):
    """The reference's SHIPPED Diamond POD model (examples/diamond/pod_model.pkl: U 4884 x 36 at tolerance 5e-5, q_ref,
    v_ref) and rest state (examples/diamond/rest.pkl) through the imported reference POD class: projections, lifts and
    model reductions of seeded full-order data (the tests regenerate the inputs from the seeds; the fixture holds the
    reference's data files and the reference's results)."""
    from _ref_import import REF
    shipped = rutils.load_data(os.path.join(REF, 'examples/diamond/pod_model.pkl'))
    rest = rutils.load_data(os.path.join(REF, 'examples/diamond/rest.pkl'))
    info = shipped['POD_info']
    U, q_ref, v_ref = np.asarray(info['U']), np.asarray(info['q_ref']), np.asarray(info['v_ref'])
    rom = rpod.POD(dict(U=U, q_ref=q_ref, v_ref=v_ref))
    n_f, r = U.shape
    rng = np.random.default_rng(180)
    Xq = q_ref + 5 * rng.standard_normal((5, n_f))
    Xv = rng.standard_normal((5, n_f))
    res = dict(U=U, q_ref=q_ref, v_ref=v_ref, rest=np.asarray(rest['rest']), seed=180)
    res['proj_rest'] = rom.compute_RO_state(qf=res['rest'])
    res['proj_q'] = np.stack([rom.compute_RO_state(qf=x) for x in Xq])
    res['proj_x'] = np.stack([rom.compute_RO_state(xf=x) for x in rutils.qv2x(Xq, Xv)])
    res['lift_x'] = np.stack([rom.compute_FO_state(x=p) for p in res['proj_x']])
    M = np.random.default_rng(181).standard_normal((n_f, n_f))
    res['UMU'] = rom.compute_RO_matrix(M)
    res['MU_rows'] = rom.compute_RO_matrix(M, right=True)[:64]
    Hm = np.random.default_rng(182).standard_normal((n_f, 4))
    res['UH'] = rom.compute_RO_matrix(Hm, left=True)
    np.savez_compressed(os.path.join(out, 'g18_pod_shipped.npz'), **res)


def ref_locp_values(case, pts, warm_start):
    """Instantiate the REFERENCE `LOCP` (sofacontrol/scp/locp.py, executed through the evaluating cvxpy stand-in),
    `update` it with the case data and evaluate its own objective (locp.py:218-263) and every constraint's residual
    (locp.py:265-342, in the order the reference appends them) at the points.  Returns J (len(pts),) and the
    residuals (len(pts) x rows)."""
    from sofacontrol.scp import locp as rlocp
    poly = lambda t: None if t is None else rutils.Polyhedron(A=np.asarray(t[0]), b=np.asarray(t[1]))
    tr = case.get('tr_active', True)
    nl = 'Hd' in case
    kw = {}
    if not tr:
        kw['is_tr_active'] = False
    if nl:
        kw['nonlinear_observer'] = True
    if case.get('input_nullspace') is not None:
        kw['input_nullspace'] = case['input_nullspace']
    (lo, _) = quiet(rlocp.LOCP, case['N'], case['H'], case['Qz'], case['R'], Qzf=case.get('Qzf'), U=poly(case.get('U')),
                    X=poly(case.get('X')), Xf=poly(case.get('Xf')), dU=poly(case.get('dU')), verbose=False,
                    warm_start=warm_start, x_char=1.0 / case['x_scale'], **kw)
    ukw = dict(z=case.get('z'), zf=case.get('zf'), u=case.get('u_des'))
    if nl:
        ukw.update(Hd=list(case['Hd']), cd=list(case['cd']))
    quiet(lo.update, list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'] if tr else None,
          case['delta'] if tr else 0.0, case['omega'] if tr else 0.0, **ukw)
    Js, Rs = [], []
    for (x, u, s) in pts:
        lo.x.value = x.ravel()
        lo.u.value = u.ravel()
        if tr:
            lo.st.value = s
        Js.append(lo.prob.objective.value)
        Rs.append(np.concatenate([c.residual() for c in lo.prob.constraints]))
    return np.array(Js), np.stack(Rs)


def g14_locp(out):
    """The QP statement of the reference's own locp.py at seeded points and at the oracle optimum, for every entry of
    qp_cases.G14_CASES (trust region on/off, U, X, Xf, dU, u_des, Qzf / zf, nonlinear observer), both through the
    Parameter path (warm_start=True) and the rebuild path (warm_start=False)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import qp_cases
    res = {}
    for name in qp_cases.G14_CASES:
        case = qp_cases.g14_case(name)
        pts = qp_cases.g14_points(name, case)
        # plus the oracle's optimum (data: stored, so that the test does not have to solve)
        kw = dict(case)
        qp = olocp.build_qp(kw.pop('N'), kw.pop('H'), kw.pop('Qz'), kw.pop('R'), kw.pop('Ad'), kw.pop('Bd'), kw.pop('dd'),
                            kw.pop('x0'), kw.pop('xk'), kw.pop('delta'), kw.pop('omega'), **kw)
        w, _, info = olocp.solve_exact(qp)
        assert info['status'] == 'optimal', (name, info)
        xo, uo, so = olocp.split(qp, w)
        pts.append((xo, uo, so))
        J1, R1 = ref_locp_values(case, pts, True)
        if 'Hd' not in case:
            # both paths of locp.py state the same QP.  (Not with a nonlinear observer: the rebuild path flattens the
            # (N+1, n_z) array `cd` with cvxpy's column-major reshape, locp.py:232-233, while the Parameter path gets
            # np.ravel(cd), locp.py:131-132 -- the default warm_start=True path is the one recorded and mirrored.)
            J2, R2 = ref_locp_values(case, pts, False)
            assert np.array_equal(J1, J2) and np.array_equal(R1, R2), name
        res[name + '_J'], res[name + '_res'], res[name + '_wopt'] = J1, R1, w
    np.savez_compressed(os.path.join(out, 'g14_locp.npz'), **res)


def g21_locp_nullspace(out):
    """The reference's objective WITH its input_nullspace term (locp.py:70-71, 258-261), evaluated by the reference's own locp.py
    (through the evaluating cvxpy stand-in) at seeded points and at the oracle's optimum, for qp_cases.NULLSPACE_CASES: a vector
    (|sum_k v . u_k|) and a matrix (|| M sum_k u_k ||_2), each with the optimum where the norm is smooth and inside its kink."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import qp_cases
    res = {}
    for name in qp_cases.NULLSPACE_CASES:
        case, ns = qp_cases.nullspace_case(name)
        pts = qp_cases.g14_points('nullspace_' + name, case)
        kw = dict(case)
        qp = olocp.build_qp(kw.pop('N'), kw.pop('H'), kw.pop('Qz'), kw.pop('R'), kw.pop('Ad'), kw.pop('Bd'), kw.pop('dd'),
                            kw.pop('x0'), kw.pop('xk'), kw.pop('delta'), kw.pop('omega'), **kw)
        w, J, info = olocp.solve_with_nullspace(qp, ns)
        cert = olocp.nullspace_certificate(qp, ns, w, info['mu'])
        assert cert['mu_norm'] <= 1 + 1e-9 and abs(cert['gap']) <= 1e-9 and abs(cert['inner_dJ']) <= 1e-7, (name, cert)
        pts.append(olocp.split(qp, w))
        J1, R1 = ref_locp_values(dict(case, input_nullspace=ns), pts, True)
        J2, R2 = ref_locp_values(dict(case, input_nullspace=ns), pts, False)
        assert np.array_equal(J1, J2) and np.array_equal(R1, R2), name
        res[name + '_J'], res[name + '_res'], res[name + '_wopt'], res[name + '_mu'] = J1, R1, w, np.atleast_1d(info['mu'])
        res[name + '_Jopt'] = np.array(J)
    np.savez_compressed(os.path.join(out, 'g21_locp_nullspace.npz'), **res)


def g19_preprocess(out):
    """process_snapshots / compute_kmeans_centroids of the imported reference (pod.py:157-178, 207-216; sklearn is a
    dependency of the reference and is installed in the build container) on small seeded snapshot sets."""
    rng = np.random.default_rng(19)
    res = {}
    # three well separated blobs + a structureless cloud
    blobs = np.concatenate([c + 0.3 * rng.standard_normal((20, 37)) for c in (np.zeros(37), 4.0 * np.ones(37), -3.0 * np.arange(37) / 37)])
    cloud = rng.standard_normal((90, 23)) * (1.0 + np.arange(23))
    for name, S, k in (('blobs', blobs, 3), ('cloud', cloud, 7)):
        res[name] = S
        res[name + '_normalize'] = rpod.process_snapshots(S.copy(), ['normalize'], {})
        res[name + '_mean'] = rpod.process_snapshots(S.copy(), ['substract_mean'], {})
        res[name + '_both'] = rpod.process_snapshots(S.copy(), ['normalize', 'substract_mean'], {})
        res[name + '_k'] = k
        res[name + '_centroids'] = rpod.process_snapshots(S.copy(), ['clustering'], dict(nbr_clusters=k))
        res[name + '_all'] = rpod.process_snapshots(S.copy(), ['normalize', 'substract_mean', 'clustering'], dict(nbr_clusters=k))
    np.savez_compressed(os.path.join(out, 'g19_preprocess.npz'), **res)


GENERATORS = dict(g19_preprocess=g19_preprocess, g20_ilqr_switches=g20_ilqr_switches, g1_pod=g1_pod, g3_tpwl=g3_tpwl, g4_riccati=g4_riccati, g6_gusto=g6_gusto, g8_controllers=g8_controllers,
                  g9_ekf=g9_ekf, g10_ssm=g10_ssm, g11_ilqr_ssm=g11_ilqr_ssm, g12_assembly=g12_assembly,
                  g13_controllers2=g13_controllers2, g14_locp=g14_locp, g21_locp_nullspace=g21_locp_nullspace, g15_ssm_controllers=g15_ssm_controllers,
                  g16_dubins=g16_dubins, g17_ssm_hardware=g17_ssm_hardware, g18_pod_shipped=g18_pod_shipped)

if __name__ == '__main__':
    # one command regenerates every fixture; `make_golden.py g6_gusto g14_locp` only the named ones
    for name in (sys.argv[1:] or list(GENERATORS)):
        GENERATORS[name](HERE)
        print('%-18s %8d bytes' % (name + '.npz', os.path.getsize(os.path.join(HERE, name + '.npz'))))
