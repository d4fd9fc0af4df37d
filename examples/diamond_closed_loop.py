"""Closed loop on the Diamond shape without SOFA: the reference's driver (examples/diamond/diamond.py:262-316 --
TPWLGuSTO + GuSTOSolverNode + the `scp` controller of tpwl/controllers.py) on a synthetic TPWL model, with the TPWL
model itself as the plant.  Every simulation step goes through the same calls SOFA's ClosedLoopController makes:
full-order state -> `controller.evaluate(sim_time, y, x, u_prev)` -> POD projection -> observer -> (re)plan with
the fused GuSTO kernel -> LQR feedback input.

    python examples/diamond_closed_loop.py [--ekf] [--steps 200]

Needs an MI355X (no CPU fallback)."""
import argparse
import contextlib
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'soft-robot-control_amd'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=200, help='simulation steps of 0.01 s')
    ap.add_argument('--ekf', action='store_true', help='DiscreteEKFObserver on 10 measured nodes instead of full state')
    args = ap.parse_args()

    import workloads as wl
    from sofacontrol_amd.measurement_models import linearModel
    from sofacontrol_amd.tpwl.tpwl import TPWLATV
    from sofacontrol_amd.tpwl import controllers as ctl
    from sofacontrol_amd.tpwl.observer import DiscreteEKFObserver
    from sofacontrol_amd.scp.models.tpwl import TPWLGuSTO
    from sofacontrol_amd.scp.standalone import GuSTOSolverNode
    from sofacontrol_amd.utils import QuadraticCost, HyperRectangle, Polyhedron

    w = wl.diamond_c2()
    n_f, r = w['U'].shape
    m, N, dt_plan, dt_sim = w['m'], w['N'], w['dt'], 0.01
    tip = 1354
    num_nodes = n_f // 3
    Hf = linearModel(nodes=[tip], num_nodes=num_nodes).C.tocsr()             # tip velocity + position (diamond.py:269)
    Cf = linearModel(nodes=list(range(0, 1500, 150)), num_nodes=num_nodes, vel=False).C.tocsr() if args.ekf else None
    data = dict(w['tab'], rom_info=dict(type='POD', U=w['U'], q_ref=w['q_ref'], v_ref=w['v_ref']))
    params = dict(tpwl_method='nn', dist_weights={'q': 1.0, 'v': 0.0})
    model = TPWLATV(data=data, params=params, Hf=Hf, Cf=Cf, discr_method='zoh')
    plant = TPWLATV(data=data, params=params, Hf=Hf, discr_method='zoh')       # its own device tables (dt_sim)
    gm = TPWLGuSTO(model)
    quiet = contextlib.redirect_stdout(io.StringIO())
    with quiet:
        gm.pre_discretize(dt_plan)
        plant.pre_discretize(dt_sim)
        x0 = np.zeros(2 * r)
        node = GuSTOSolverNode(gm, N, dt_plan, w['Qz'], w['R'], x0, t=w['t'], z=w['z'], U=HyperRectangle([1500.] * m, [0.] * m),
                               X=Polyhedron(w['XA'], w['Xb']), convg_thresh=1e-3, max_gusto_iters=3)
        H = np.asarray(model.H)
        cost = QuadraticCost(Q=H.T @ w['Qz'] @ H + 1e-3 * np.eye(2 * r), R=1e-4 * np.eye(m))
        obs = DiscreteEKFObserver(model, W=10 * np.eye(2 * r), V=1e-2 * np.eye(30)) if args.ekf else None
        c = ctl.scp(model, cost, dt_sim, N_replan=10, observer=obs, delay=0.0, solver_node=node)
        c.set_sim_timestep(dt_sim)
    V = np.kron(np.eye(2), w['U'])
    x_ref = np.concatenate((w['v_ref'], w['q_ref']))
    zi = lambda t: np.array([np.interp(t, w['t'], w['z'][:, j]) for j in range(6)])
    xr = x0.copy()
    u = np.zeros(m)
    err, t_eval, sat = [], [], []
    for k in range(args.steps):
        t = k * dt_sim
        x_full = V @ xr + x_ref                                   # what SOFA hands to the controller
        y = np.asarray(Cf @ x_full).ravel() if args.ekf else None
        t0 = time.perf_counter()
        with quiet:
            u = c.evaluate(t, y, x_full, u)
        t_eval.append(time.perf_counter() - t0)
        xr = plant.update_state(xr, u, dt_sim)                    # plant step (device gather + host affine update)
        z = H @ xr
        err.append(np.linalg.norm((z - zi(t + dt_sim))[3:5]))
        sat.append(bool(np.any(u <= 1e-6) or np.any(u >= 1500.0 - 1e-6)))
    t_eval = np.array(t_eval) * 1e3
    replans = len(c.save_controller_info()['solve_times'])
    print('steps %d (%.2f s), observer %s, %d GuSTO replans' % (args.steps, args.steps * dt_sim, 'EKF' if args.ekf else 'full state', replans))
    print('tip tracking error (x, y): rms %.3f, max %.3f   (target amplitude %.1f)' %
          (np.sqrt(np.mean(np.square(err))), np.max(err), np.abs(w['z'][:, 3:5]).max()))
    err_a, sat_a = np.array(err), np.array(sat)
    per_s = int(round(1.0 / dt_sim))
    print('per second: rms error ' + ' '.join('%.2f' % np.sqrt(np.mean(err_a[i:i + per_s] ** 2)) for i in range(0, len(err_a), per_s)))
    print('            input at a bound on ' + ' '.join('%3.0f%%' % (100 * sat_a[i:i + per_s].mean()) for i in range(0, len(sat_a), per_s)) + ' of the steps (cables only pull: U = [0, 1500]^4)')
    print('controller.evaluate per step: median %.2f ms, max %.1f ms (replan steps include the SCP solve)' %
          (np.median(t_eval), t_eval.max()))


if __name__ == '__main__':
    main()
